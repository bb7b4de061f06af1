"""``FusedAdam`` -- torch.optim.Adam semantics (reference ``src/models/train.py:461``,
``src/models/adversarial_trainer.py:56-59,191``: lr only, default betas/eps, no weight decay, no amsgrad) as ONE
HBM-bound kernel launch over a network's flat parameter arena (csrc/optim.hip).

Stock ``torch.optim.Adam`` keeps working on the same parameters (drop-in); this class is what the build's trainers use.

State layout.  The moments live in two flat buffers that mirror the parameter arena (same offsets, same channel padding),
so the update is one pass of 28 B per parameter.  What ``self.state`` / ``state_dict()`` show is torch.optim.Adam's own
format -- per parameter ``step``, ``exp_avg``, ``exp_avg_sq`` in the parameter's logical shape (strided views of the flat
buffers; ``state_dict()`` hands out dense copies) -- so ``optimizer_state_dict`` of a checkpoint (reference
``train.py:495``) loads into either optimizer, and a resumed FusedAdam continues with its moments instead of restarting
them.  When the arena is rebuilt (``.to()``, ``set_compute_dtype``) the moments are carried over through those views.

The flat pass is only taken when the parameters this optimizer was given (and that received a gradient) are EXACTLY the
parameters of the arena: a frozen encoder or an optimizer over ``model.decoder.parameters()`` takes the per-tensor path, so
nothing the caller did not hand over is ever updated.
"""
import torch

from . import kernels as K
from .engine import arena_owner


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._flat = {}                 # arena storage pointer -> {"m", "v", "step" (shared 0-dim tensor), "ids"}
        self._rebind = True             # per-parameter state may not point into the flat buffers (fresh / just loaded)
        self.flat_launches = 0          # arena-wide kernel launches of the last step() (tests)

    # ------------------------------------------------------------------------------------------------ flat arenas
    @staticmethod
    def _flat_view(part):
        """``part``: parameters (all with gradients) that share one storage.  If that storage is a network's parameter arena,
        the parameters are exactly the arena's, and the gradients sit in a mirror arena at matching offsets, return the two
        flat tensors; else None."""
        p0, g0 = part[0], part[0].grad
        sp, sg = p0.untyped_storage(), g0.untyped_storage()
        owner = arena_owner(sp.data_ptr())
        if owner is None or sp.nbytes() != sg.nbytes() or sp.nbytes() % 16:
            return None
        if len(part) != len(owner._param_list) or {id(p) for p in part} != {id(p) for p in owner._param_list}:
            return None                 # a subset (frozen / partial optimizer): never touch what was not handed over
        bp, bg = sp.data_ptr(), sg.data_ptr()
        for p in part:
            g = p.grad
            if (p.dtype != torch.float32 or g.dtype != torch.float32 or g.untyped_storage().data_ptr() != bg
                    or p.data_ptr() - bp != g.data_ptr() - bg):
                return None
        n = sp.nbytes() // 4
        flat_p = torch.empty(0, device=p0.device, dtype=torch.float32).set_(sp, 0, (n,), (1,))
        flat_g = torch.empty(0, device=p0.device, dtype=torch.float32).set_(sg, 0, (n,), (1,))
        return flat_p, flat_g

    def _bind(self, fp, part):
        """Flat moment buffers for the arena ``fp`` covers, with every parameter's ``exp_avg`` / ``exp_avg_sq`` / ``step``
        state pointing into them.  Existing per-parameter state (a loaded checkpoint, the buffers of a previous arena) is
        copied in, never dropped."""
        key = fp.data_ptr()
        fl = self._flat.get(key)
        ids = tuple(id(p) for p in part)
        if fl is not None and not self._rebind and fl["m"].numel() == fp.numel() and fl["ids"] == ids:
            return fl
        if fl is None or fl["m"].numel() != fp.numel() or fl["m"].device != fp.device:
            fl = {"m": torch.zeros_like(fp), "v": torch.zeros_like(fp), "step": None}
        steps = []
        for p in part:
            st = self.state[p]
            mv = fl["m"].as_strided(p.shape, p.stride(), p.storage_offset())
            vv = fl["v"].as_strided(p.shape, p.stride(), p.storage_offset())
            if "exp_avg" in st and st["exp_avg"].data_ptr() != mv.data_ptr():
                mv.copy_(st["exp_avg"])
                vv.copy_(st["exp_avg_sq"])
            if "step" in st:
                steps.append(float(st["step"]))
            st["exp_avg"], st["exp_avg_sq"] = mv, vv
        if steps and min(steps) != max(steps):
            raise RuntimeError("FusedAdam: parameters of one arena carry different step counts; use torch.optim.Adam")
        fl["step"] = torch.tensor(steps[0] if steps else 0.0, dtype=torch.float32)
        for p in part:
            self.state[p]["step"] = fl["step"]           # one shared counter, bumped once per launch
        fl["ids"] = ids
        # drop buffers of arenas that no longer exist (their state was carried over through the views above)
        self._flat = {k: v for k, v in self._flat.items() if k == key or arena_owner(k) is not None}
        self._flat[key] = fl
        return fl

    # ------------------------------------------------------------------------------------------------- checkpoints
    def state_dict(self):
        """torch.optim.Adam's format: dense per-parameter ``exp_avg`` / ``exp_avg_sq`` (logical shapes) and ``step``."""
        sd = super().state_dict()
        for st in sd["state"].values():
            for k, v in list(st.items()):
                if torch.is_tensor(v):
                    st[k] = v.detach().clone().contiguous() if v.dim() else v.detach().clone()
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._rebind = True              # next step() scatters the loaded moments into the flat buffers

    # -------------------------------------------------------------------------------------------------------- step
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.flat_launches = 0
        rebind_next = False
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            lr, (b1, b2), eps = group["lr"], group["betas"], group["eps"]
            # one launch per parameter arena: a group may span several networks (e.g. segmenter + feature discriminator)
            parts = {}
            for p in ps:
                parts.setdefault(p.untyped_storage().data_ptr(), []).append(p)
            rest = []
            for part in parts.values():
                flat = self._flat_view(part) if part[0].is_cuda else None
                if flat is None:
                    rest.extend(part)
                    continue
                fp, fg = flat
                fl = self._bind(fp, part)
                fl["step"] += 1
                t = int(fl["step"])
                K.adam_flat(fp, fg, fl["m"], fl["v"], fp.numel(), lr, b1, b2, eps, 1 - b1 ** t, 1 - b2 ** t)
                self.flat_launches += 1
            # parameters that are not a whole arena: same arithmetic, per tensor (dense tensors -> the kernel)
            if rest and not getattr(self, "_warned_per_tensor", False):
                import warnings
                warnings.warn(f"FusedAdam: {len(rest)} parameter(s) of this group are not a whole parameter arena (frozen layers, a "
                              "sub-module's parameters, several groups over one network): they take the per-tensor path -- same "
                              "arithmetic, one launch per tensor instead of one per network", stacklevel=2)
                self._warned_per_tensor = True
            for p in rest:
                st = self.state[p]
                if "exp_avg" not in st:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                elif any(st["step"] is fl["step"] for fl in self._flat.values()):
                    st["step"] = st["step"].clone()      # was bound to an arena-wide counter: detach it
                    rebind_next = True                   # a later whole-arena pass re-collects (and checks) the counters
                st["step"] += 1
                t = int(st["step"])
                g, m, v = p.grad, st["exp_avg"], st["exp_avg_sq"]
                dense = (p.is_cuda and p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()
                         and p.data_ptr() % 16 == 0 and g.data_ptr() % 16 == 0 and m.data_ptr() % 16 == 0
                         and v.data_ptr() % 16 == 0 and p.dtype == torch.float32)
                if dense:
                    K.adam_flat(p, g, m, v, p.numel(), lr, b1, b2, eps, 1 - b1 ** t, 1 - b2 ** t)
                else:
                    m.lerp_(g, 1 - b1)
                    v.mul_(b2).addcmul_(g, g, value=1 - b2)
                    denom = (v.sqrt() / (1 - b2 ** t) ** 0.5).add_(eps)
                    p.addcdiv_(m, denom, value=-lr / (1 - b1 ** t))
        self._rebind = rebind_next
        return loss
