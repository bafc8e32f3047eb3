"""Model preparation for TTA — mirrors dg_tta/tta/model_utils.py (get_model_from_network :12-35, running-stat
buffering :41-63, which is a no-op for InstanceNorm without running statistics)."""
from copy import deepcopy

from .torch_utils import register_forward_pre_hook_at_beginning, register_forward_hook_at_beginning, hookify


def get_model_from_network(network, modifier_fn_module, parameters=None):
    model = deepcopy(network)
    if parameters is not None:
        target = getattr(model, "_orig_mod", model)
        target.load_state_dict(parameters[0])
    mf = modifier_fn_module.ModifierFunctions
    register_forward_pre_hook_at_beginning(model, hookify(mf.modify_tta_input_fn, "forward_pre_hook"))
    register_forward_hook_at_beginning(model, hookify(mf.modfify_tta_model_output_fn, "forward_hook"))
    return model


running_stats_buffer = {}


def buffer_running_stats(m):
    if getattr(m, "running_mean", None) is not None and getattr(m, "running_var", None) is not None \
            and id(m) not in running_stats_buffer:
        running_stats_buffer[id(m)] = [m.running_mean.data, m.running_var.data]


def apply_running_stats(m):
    if hasattr(m, "running_mean") and hasattr(m, "running_var") and id(m) in running_stats_buffer:
        mean, var = running_stats_buffer.pop(id(m))
        m.running_mean.data.copy_(mean)
        m.running_var.data.copy_(var)
