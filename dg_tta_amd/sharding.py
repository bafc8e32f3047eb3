"""Sample sharding for multi-GPU TTA: units are independent (reference: tta.py:157-182), so ranks only need a static
partition and — for benchmarking — a barrier and a max-over-ranks reduction of the elapsed time.  No data-path collective."""
import os


def rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def owns(sample_index, rank, world):
    return sample_index % world == rank


def unit_owner(sample_index, ensemble_index, num_samples, ensemble_count, world):
    """Rank that adapts the (sample, ensemble member) unit.  With at least as many samples as ranks a sample's members
    stay together (round-robin over samples); with fewer samples than ranks the (sample, member) PAIRS are dealt
    round-robin, so that e.g. 1 sample x 3 members still uses 3 GPUs.  The sample's ensemble inference
    is done by the owner of member 0, after a filesystem barrier on the members' parameter files."""
    if world <= 1:
        return 0
    if num_samples >= world:
        return sample_index % world
    return (sample_index * ensemble_count + ensemble_index) % world


def units_for_rank(num_samples, ensemble_count, rank, world):
    """(sample, ensemble) units of this rank (see unit_owner)."""
    return [(s, e) for s in range(num_samples) for e in range(ensemble_count)
            if unit_owner(s, e, num_samples, ensemble_count, world) == rank]


def wait_for_files(paths, timeout_s=6 * 3600.0, poll_s=0.5, abort_if=()):
    """Filesystem barrier (the mechanism the reference's resume logic already relies on, tta.py:164-170): blocks until
    every path exists; raises TimeoutError naming the missing files, or RuntimeError as soon as one of the `abort_if`
    paths (the peers' failure markers) appears.  Writers rename complete files into place."""
    import time
    from pathlib import Path
    paths = [Path(p) for p in paths]
    abort_if = [Path(p) for p in abort_if]
    t0 = time.monotonic()
    while True:
        missing = [p for p in paths if not p.is_file()]
        if not missing:
            return
        dead = [p for p in abort_if if _marker_of_this_launch(p, "failed")]
        if dead:
            raise RuntimeError(f"filesystem barrier: a peer rank failed ({[p.name for p in dead]}); still missing "
                               f"{[str(p) for p in missing]}")
        if time.monotonic() - t0 > timeout_s:
            raise TimeoutError(f"filesystem barrier: still missing after {timeout_s:.0f} s: {[str(p) for p in missing]}")
        time.sleep(poll_s)


def _marker_of_this_launch(path, word):
    """True if `path` exists and was written by THIS launch (markers carry the launch id; one of an earlier launch in a
    resumed run directory is ignored).  A marker that vanishes between the check and the read (its owner unlinks stale
    ones at start-up) counts as absent."""
    try:
        return path.read_text() == f"{word} {launch_id()}\n"
    except (FileNotFoundError, NotADirectoryError):
        return False


def mark_rank_failed(save_path, rank):
    try:
        failed_marker(save_path, rank).write_text(f"failed {launch_id()}\n")
    except OSError:
        pass            # (never mask the original error)


def done_marker(save_path, rank):
    from pathlib import Path
    return Path(save_path) / f".rank_{rank}.done"


def failed_marker(save_path, rank):
    from pathlib import Path
    return Path(save_path) / f".rank_{rank}.failed"


def _parent_identity():
    """(pid, start time in clock ticks since boot) of the parent process: the same for every rank torch.distributed.run's
    agent spawned, different for every launch (a recycled pid cannot have the same start tick)."""
    ppid = os.getppid()
    try:
        with open(f"/proc/{ppid}/stat") as f:
            start = f.read().rsplit(")", 1)[1].split()[19]          # field 22 (starttime); the comm field may hold spaces
    except (OSError, IndexError):
        start = "?"
    return f"{ppid}.{start}"


def launch_id():
    """Identifies ONE launch of a multi-rank run; written into the done / failed markers so that a resumed run in the same
    directory neither passes the barrier nor aborts on a previous launch's markers.  `dgtta run_tta --gpus N` hands its
    children a fresh DGTTA_LAUNCH_ID.  Under torch.distributed.run:

    * a real rendezvous id (`--rdzv-id X`, anything but the constant 'none') is common to every node of the job:
      the id is X plus TORCHELASTIC_RESTART_COUNT, so that the markers of a failed attempt do not count for the retry;
    * the constant 'none' on ONE node (LOCAL_WORLD_SIZE == WORLD_SIZE): the identity (pid, start tick) of the elastic
      agent - the common parent of all ranks of the launch - tells two launches apart; the restart count is added too;
    * the constant 'none' across SEVERAL nodes identifies nothing (each node's agent is another process): refused -
      pass --rdzv-id or export DGTTA_LAUNCH_ID.

    Any other external launcher has to export DGTTA_LAUNCH_ID itself (`dgtta run_tta` refuses WORLD_SIZE > 1 without
    one); empty only for in-process use (tests that run the ranks one after the other)."""
    lid = os.environ.get("DGTTA_LAUNCH_ID")
    if lid:
        return lid
    run_id = os.environ.get("TORCHELASTIC_RUN_ID")
    if run_id is None:
        return ""
    attempt = os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
    if run_id != "none":
        return f"{run_id}#{attempt}"
    world = os.environ.get("WORLD_SIZE", "1")
    if os.environ.get("LOCAL_WORLD_SIZE", world) != world:
        raise RuntimeError("dg_tta_amd.sharding.launch_id: a multi-node torch.distributed.run job without --rdzv-id "
                           "(TORCHELASTIC_RUN_ID is the constant 'none') has no id common to its nodes; pass "
                           "--rdzv-id <unique string> or export DGTTA_LAUNCH_ID (same on every rank, new per launch)")
    return f"none@{_parent_identity()}#{attempt}"


def mark_rank_done(save_path, rank):
    m = done_marker(save_path, rank)
    tmp = m.with_name(m.name + ".tmp")
    tmp.write_text(f"done {launch_id()}\n")
    tmp.replace(m)


def wait_for_done_markers(save_path, world, timeout_s, poll_s=0.5):
    """Barrier of the summary: every rank's marker exists AND belongs to this launch."""
    import time
    t0 = time.monotonic()
    while True:
        wait_for_files([done_marker(save_path, r) for r in range(world)], max(timeout_s - (time.monotonic() - t0), 0.0),
                       poll_s, abort_if=[failed_marker(save_path, r) for r in range(world)])
        stale = [r for r in range(world) if not _marker_of_this_launch(done_marker(save_path, r), "done")]
        if not stale:
            return
        if time.monotonic() - t0 > timeout_s:
            raise TimeoutError(f"filesystem barrier: ranks {stale} only left markers of an earlier launch in {save_path}")
        time.sleep(poll_s)


def summary_by_parent():
    """True in the children of `dgtta run_tta --gpus N`: the fan-out parent evaluates the run after all of them exited."""
    return os.environ.get("DGTTA_SUMMARY_BY_PARENT", "0") == "1"


def child_devices(n):
    """Physical device ids for n children: the parent's HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES restriction is kept."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        cur = os.environ.get(var)
        if cur:
            ids = [c.strip() for c in cur.split(",") if c.strip() != ""]
            if len(ids) < n:
                raise RuntimeError(f"--gpus {n} but {var}={cur} exposes only {len(ids)} device(s)")
            return ids[:n]
    return [str(i) for i in range(n)]


def max_over_ranks(seconds, device=None):
    """Slowest rank's time (what a whole-job throughput must be quoted on)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
