"""Process-level HIP runtime settings this package depends on.  Imported first by tinyedm_amd/__init__.py.

hipGraph replay on ROCm 7.2 (measured round 2): with the runtime's default "AQL packet capture"
fast path, the first replay of an instantiated graph that follows a hipStreamSynchronize / hipDeviceSynchronize
runs some of its nodes with clobbered kernel arguments -- a captured training step then turns its weights into
garbage / NaN (and, drawing less power, runs faster: the bug first showed up as a "too good" bench number).  Event
synchronisation does not trigger it; `DEBUG_CLR_GRAPH_PACKET_CAPTURE=0` removes it at no measurable replay cost.  The
runtime reads the variable at its initialisation (the first HIP call).

FAIL CLOSED: the graph paths of this package run only when the safe value provably reached the runtime, i.e. when
  (a) the variable was already "0" in the environment this process inherited, or
  (b) the HIP runtime was not initialised yet when this module set it -- judged by whether the process holds /dev/kfd
      open (the runtime's initialisation opens it; torch's own lazy-init flag is NOT a witness: torch.cuda.is_available()
      initialises HIP without setting it).
Anything else -- including a value set by this module after the runtime came up -- reports unsafe: the trainer and
`generate` fall back to the eager loop, CapturedTrainStep / solve(graph=True) raise.  Independently of this switch
every graph path carries an in-graph finiteness sentinel (ops.health / ops.check_health), so a corrupted replay fails
loudly instead of silently."""
import os

VAR = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"


def hip_runtime_live() -> bool:
    """True when this process has initialised the HIP/HSA runtime (it holds the KFD device node open)."""
    try:
        for fd in os.listdir("/proc/self/fd"):
            try:
                if os.readlink(f"/proc/self/fd/{fd}") == "/dev/kfd":
                    return True
            except OSError:
                continue
    except OSError:
        return True        # cannot tell: assume the worst
    return False


def _prepare() -> bool:
    cur = os.environ.get(VAR)
    live = hip_runtime_live()
    if cur is not None:
        # inherited from the launcher: safe iff it is the safe value (a "0" written into os.environ by this very process
        # after the runtime came up does not count -- nothing but this module writes it, and it does so only when not live)
        return cur == "0"
    if live:
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
