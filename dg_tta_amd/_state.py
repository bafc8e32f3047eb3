"""Per-model host state of the TTA engine.

Everything the host layer has to remember between two calls lives here, keyed by the MODEL object: the MIND noise /
descriptor handed from the input-preparation step to the model's `mind_hook`, the per-window grouping the sliding-window
inference asks for, the side streams of the backward pass and of the input pipeline, and bench.py's launch probe.
Two TTA instances in one process (two threads, two streams) therefore never see each other's hand-overs or streams.
Entries vanish with their model (weak keys); a deep copy of a model (`get_model_from_network`) starts with fresh state.
"""
import threading
import weakref


class ModelState:
    __slots__ = ("noise", "features", "forced_groups", "side_stream", "prep_stream", "probe", "__weakref__")

    def __init__(self):
        self.noise = []            # [(noise tensor, groups)] drawn ahead of the network pass (reference draw order)
        self.features = []         # MIND descriptors evaluated ahead of the network pass
        self.forced_groups = None  # mind_groups(): the batch is this many independent MIND calls
        self.side_stream = None    # weight-gradient stream of the backward pass
        self.prep_stream = None    # input pipeline (patch sampling, GIN, warp, MIND) of the next pass
        self.probe = None          # bench.py: events around one block's forward conv launch

    def stream(self, which, device):
        """Lazily created side stream `which` in {'side_stream', 'prep_stream'} on `device`."""
        import torch
        st = getattr(self, which)
        d = torch.device(device)
        idx = d.index if d.index is not None else torch.cuda.current_device()
        if st is None or st.device.index != idx:
            st = torch.cuda.Stream(device=d)
            setattr(self, which, st)
        return st


_STATES = weakref.WeakKeyDictionary()
_LOCK = threading.Lock()


def state_of(model):
    with _LOCK:
        st = _STATES.get(model)
        if st is None:
            st = _STATES[model] = ModelState()
        return st
