"""Config round-trip helpers (reference utils.py:5-34)."""
import inspect

import torch


def deinstantiate(obj):
    """Module -> Hydra-instantiable dict: ``_target_`` + every constructor argument read back from the
    same-named attribute (nested modules recursively, tuples as lists) -- reference utils.py:5-27.
    ``_target_`` uses the *tinyedm* alias path so checkpoints stay interchangeable with the reference."""
    mod = obj.__class__.__module__.replace("tinyedm_amd", "tinyedm", 1)
    target = f"{mod}.{obj.__class__.__name__}"
    out = {}
    for name, param in inspect.signature(obj.__init__).parameters.items():
        if name == "self":
            continue
        if hasattr(obj, name):
            value = getattr(obj, name)
            if isinstance(value, torch.nn.Module):
                out[name] = deinstantiate(value)
            elif isinstance(value, tuple):
                out[name] = list(value)
            else:
                out[name] = value
        elif param.default is not inspect.Parameter.empty:
            out[name] = param.default
    return {"_target_": target, **out}


def swap_tensors(tensor1, tensor2):
    """In-place exchange of two same-shaped tensors (reference utils.py:30-34)."""
    tmp = tensor1.clone()
    tensor1.copy_(tensor2)
    tensor2.copy_(tmp)
