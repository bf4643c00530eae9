"""Process-level HIP runtime settings this package depends on.  Imported first by tinyedm_amd/__init__.py.

hipGraph replay on ROCm 7.2 (measured round 2, tools/nan_hunt.py): with the runtime's default "AQL packet capture"
fast path, the first replay of an instantiated graph that follows a hipStreamSynchronize / hipDeviceSynchronize
runs some of its nodes with clobbered kernel arguments -- a captured training step then turns its weights into
garbage / NaN (and, drawing less power, runs faster: the bug first showed up as a "too good" bench number).  Event
synchronisation does not trigger it; `DEBUG_CLR_GRAPH_PACKET_CAPTURE=0` removes it at no measurable replay cost.  The
runtime reads the variable at its initialisation (the first HIP call), so it is set here, at import time, when the
process has not initialised the GPU yet; otherwise the graph paths of this package refuse to run."""
import os
import sys

VAR = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"


def _prepare() -> bool:
    cur = os.environ.get(VAR)
    if cur is not None:
        return cur == "0"          # whoever set it before launch decided; "0" is the safe value
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_initialized():
        return False               # too late for this process: the runtime has read its flags
    os.environ[VAR] = "0"
    return True


GRAPH_REPLAY_SAFE = _prepare()


def require_graph_replay_safe(what: str):
    if not GRAPH_REPLAY_SAFE:
        raise RuntimeError(
            f"{what}: hipGraph replay is unsafe in this process -- the HIP runtime was initialised before tinyedm_amd "
            f"was imported (or {VAR} is set to a value other than 0).  Import tinyedm / tinyedm_amd before the first "
            f"GPU call, or export {VAR}=0; see tinyedm_amd/_runtime_env.py")
