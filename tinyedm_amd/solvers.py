"""2nd-order Heun sampler (reference solvers.py:4-59) with the loop optionally captured in a hipGraph."""
import weakref

import torch

from . import _runtime_env, ops


def _release_solves(per_model: dict):
    try:
        for ent in per_model.values():
            ops.release_capture(ent[5])
        per_model.clear()
    except Exception:       # noqa: BLE001  (interpreter shutdown)
        pass


class DeterministicSolver:
    """Algorithm 1 of Karras et al. 2022 with sigma(t)=t, s(t)=1.  Same constructor as the reference.

    The sigma table is built with the reference's exact fp32 expression (bitwise-equal table,
    solvers.py:33-41) and uploaded to the device ONCE: the reference's per-step ``t0.to(device)``
    host->device copies (63 sync points for 32 steps) disappear, which is what makes the whole
    solve capturable as one hipGraph (``solve(..., graph=True)``)."""

    MAX_GRAPHS = 4      # captured solves kept per model (shape / precision combinations; least recently used dropped)

    def __init__(self, num_steps: int = 18, sigma_min: float = 0.002, sigma_max: float = 80.0, rho: float = 7.0,
                 dtype: str | None = None):
        self.num_steps = num_steps
        self.sigma_min = sigma_min
        self.sigma_max = sigma_max
        self.rho = rho
        if dtype not in (None, "float32"):
            raise ValueError("tinyedm_amd.DeterministicSolver integrates in float32 (the reference's only working "
                             "setting: its .to(dtype) call crashes for string dtypes)")
        self.dtype = torch.float32
        i = torch.arange(num_steps, dtype=torch.float32)
        t = (sigma_max ** (1 / rho) + i / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
        self.t_steps = torch.cat([t, torch.zeros(1)])
        self._graphs = weakref.WeakKeyDictionary()      # model -> {(shapes, device): captured solve}

    # ------------------------------------------------------------------ eager
    def _loop(self, model, x0, class_labels, t_dev):
        ts = self.t_steps.tolist()
        x1 = ops.scale_f32(x0, ts[0])
        for i in range(self.num_steps):
            t0, t1 = ts[i], ts[i + 1]
            x = x1
            D = model(x, t_dev[i], class_labels).float().contiguous()
            dx, x1 = ops.heun_euler(x, D, t0, t1)
            if i < self.num_steps - 1:
                D1 = model(x1, t_dev[i + 1], class_labels).float().contiguous()
                x1 = ops.heun_correct(x, dx, x1, D1, t0, t1)
        return x1

    @torch.no_grad()
    def solve(self, model, x0, class_labels=None, graph: bool = False):
        if not x0.is_cuda:
            raise RuntimeError("tinyedm_amd.DeterministicSolver: x0 must be a GPU tensor (there is no CPU path)")
        in_dtype = x0.dtype
        x0 = x0.float().contiguous()
        if not graph:
            t_dev = self.t_steps.to(x0.device)
            return self._loop(model, x0, class_labels, t_dev).to(in_dtype)
        out = self._solve_graphed(model, x0, class_labels).to(in_dtype)
        # the Heun kernels leave a bit in the device health word when the state went non-finite: a replay that ran
        # with corrupted arguments fails HERE, loudly (one host sync per solve of 2N-1 network evaluations)
        ops.check_health(x0.device, "DeterministicSolver.solve(graph=True)")
        return out

    # ------------------------------------------------------------------ hipGraph
    def _solve_graphed(self, model, x0, class_labels):
        # graphs are cached PER MODEL OBJECT (weakly: a new model allocated at a dead one's address must not replay the
        # dead one's graph, which id(model) as a key allowed)
        owner = getattr(model, "__self__", model)        # a bound method is a fresh object per access: key on its object
        per_model = self._graphs.get(owner)
        if per_model is None:
            per_model = self._graphs[owner] = {}
            # when the model is collected its captured solves go with it: give their launch-table slots and plan pins back
            weakref.finalize(owner, _release_solves, per_model)
        # the evaluation precision of the denoiser(s) is part of the key: set_eval_dtype() between two solves must not
        # replay a graph captured with the other path's kernels
        dtypes = ()
        if isinstance(owner, torch.nn.Module):
            dtypes = tuple(getattr(m, "eval_dtype", None) for m in owner.modules() if hasattr(m, "eval_dtype"))
        key = (tuple(x0.shape), None if class_labels is None else tuple(class_labels.shape), x0.device.index, dtypes)
        ent = per_model.get(key)
        if ent is None:
            _runtime_env.require_graph_replay_safe("DeterministicSolver.solve(graph=True)")
            t_dev = self.t_steps.to(x0.device)
            sx = x0.clone()
            sl = None if class_labels is None else class_labels.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):          # warm-up outside capture (weight packs, lazy inits)
                self._loop(model, sx, sl, t_dev)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            ops.capture_begin()
            ok = False
            try:
                with torch.cuda.graph(g):
                    out = self._loop(model, sx, sl, t_dev)
                ok = True
            finally:
                token = ops.capture_end()
                if not ok:
                    ops.release_capture(token)
            while len(per_model) >= self.MAX_GRAPHS:         # (dict order = least recently used first)
                old = per_model.pop(next(iter(per_model)))
                torch.cuda.synchronize()
                ops.release_capture(old[5])
            ent = per_model[key] = (g, sx, sl, out, t_dev, token)
        else:
            per_model[key] = per_model.pop(key)              # most recently used last
        g, sx, sl, out, _, _ = ent
        # the captured evaluations read the persistent eval-mode weight packs: refresh them (a no-op unless the
        # master weights changed since the last solve: optimizer steps, EMA swap, load_state_dict) before replaying
        if isinstance(model, torch.nn.Module):
            from .networks import Denoiser
            for m in model.modules():
                if isinstance(m, Denoiser) and not m.training:
                    m._prep_all()
        sx.copy_(x0)
        if sl is not None:
            sl.copy_(class_labels)
        g.replay()
        return out.clone()
