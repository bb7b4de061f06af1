"""``FusedAdam`` -- torch.optim.Adam semantics (reference ``src/models/train.py:461``,
``src/models/adversarial_trainer.py:56-59,191``: lr only, default betas/eps, no weight decay, no amsgrad) as ONE
HBM-bound kernel launch over a network's flat parameter arena (csrc/optim.hip).

Stock ``torch.optim.Adam`` keeps working on the same parameters (drop-in); this class is what the build's trainers use.
"""
import torch

from . import kernels as K


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @staticmethod
    def _flat_view(ps):
        """If all params (and all grads) are views of one storage each, at matching offsets, return the two flat
        tensors covering those storages; else None."""
        p0, g0 = ps[0], ps[0].grad
        sp, sg = p0.untyped_storage(), g0.untyped_storage()
        if sp.nbytes() != sg.nbytes() or sp.nbytes() % 16:
            return None
        bp, bg = sp.data_ptr(), sg.data_ptr()
        for p in ps:
            g = p.grad
            if (p.dtype != torch.float32 or g.dtype != torch.float32 or p.untyped_storage().data_ptr() != bp
                    or g.untyped_storage().data_ptr() != bg or p.data_ptr() - bp != g.data_ptr() - bg):
                return None
        n = sp.nbytes() // 4
        flat_p = torch.empty(0, device=p0.device, dtype=torch.float32).set_(sp, 0, (n,), (1,))
        flat_g = torch.empty(0, device=p0.device, dtype=torch.float32).set_(sg, 0, (n,), (1,))
        return flat_p, flat_g

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            lr, (b1, b2), eps = group["lr"], group["betas"], group["eps"]
            # one launch per parameter arena: a group may span several networks (e.g. segmenter + feature discriminator)
            parts = {}
            for p in ps:
                parts.setdefault(p.untyped_storage().data_ptr(), []).append(p)
            rest = []
            for k, part in enumerate(parts.values()):
                flat = self._flat_view(part) if part[0].is_cuda else None
                if flat is None:
                    rest.extend(part)
                    continue
                fp, fg = flat
                st = self.state.setdefault(("flat", gi) if k == 0 else ("flat", gi, k), {})
                if "m" not in st or st["m"].numel() != fp.numel() or st["ptr"] != fp.data_ptr():
                    st["m"], st["v"] = torch.zeros_like(fp), torch.zeros_like(fp)
                    st["step"], st["ptr"] = st.get("step", 0), fp.data_ptr()
                st["step"] += 1
                t = st["step"]
                K.adam_flat(fp, fg, st["m"], st["v"], fp.numel(), lr, b1, b2, eps, 1 - b1 ** t, 1 - b2 ** t)
            ps = rest
            # parameters that do not share an arena: same arithmetic, per tensor (dense tensors -> the kernel)
            for p in ps:
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["m"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["v"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                t = st["step"]
                g, m, v = p.grad, st["m"], st["v"]
                dense = (p.is_cuda and p.is_contiguous() and g.is_contiguous() and p.data_ptr() % 16 == 0
                         and g.data_ptr() % 16 == 0 and p.dtype == torch.float32)
                if dense:
                    K.adam_flat(p, g, m, v, p.numel(), lr, b1, b2, eps, 1 - b1 ** t, 1 - b2 ** t)
                else:
                    m.lerp_(g, 1 - b1)
                    v.mul_(b2).addcmul_(g, g, value=1 - b2)
                    denom = (v.sqrt() / (1 - b2 ** t) ** 0.5).add_(eps)
                    p.addcdiv_(m, denom, value=-lr / (1 - b1 ** t))
        return loss
