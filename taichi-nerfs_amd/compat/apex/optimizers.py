"""apex.optimizers.FusedAdam as the reference's train.py:143-149 constructs it (`FusedAdam(model.parameters(), lr=..., eps=1e-15)`),
on top of the C ABI's one-pass Adam (`ngp_adam_multi`, csrc/optim.hip: unscale + moments + update + gradient clearing in a single
sweep over p/g/m/v of every tensor of a parameter group, one launch; float4 groups never touched by a step are skipped as exact
fixed points).

What the unchanged driver gains over its torch.optim.Adam fall-back: torch's GradScaler.step() makes a separate unscale pass over
every gradient and then READS THE INF FLAG BACK to the host before it may call step() -- one host sync per iteration.  This class
declares `_step_supports_amp_scaling`: the scaler hands it `grad_scale` / `found_inf` as device tensors, the unscale happens inside
the Adam sweep and a skipped step is skipped on the device (`ngp_adam_amp_prologue`).  Same arithmetic as
torch.optim.Adam(eps=1e-15) (tests/test_gpu_apex_adam.py holds it to a 50-step torch loop incl. an overflow step).
Scope: Adam without weight decay / amsgrad on contiguous fp32 CUDA tensors whose size is a multiple of 4 (every parameter of the
reference's models); anything else raises."""
import ctypes

import torch

from ngp_hip import lib as _lib
from ngp_hip.ops import _ptr, _stream, _touched


class FusedAdam(torch.optim.Optimizer):
    _step_supports_amp_scaling = True

    def __init__(self, params, lr=1e-3, bias_correction=True, betas=(0.9, 0.999), eps=1e-8, adam_w_mode=True, weight_decay=0.0,
                 amsgrad=False, set_grad_none=True):
        if amsgrad:
            raise RuntimeError("FusedAdam does not support the AMSGrad variant.")          # (apex's own message)
        if weight_decay != 0.0:
            raise NotImplementedError("compat FusedAdam: weight_decay != 0 is not implemented (the reference uses 0)")
        if not bias_correction:
            raise NotImplementedError("compat FusedAdam: bias_correction=False is not implemented")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.set_grad_none = set_grad_none
        self._L = _lib.load()
        self._sf = self._si = None
        self._multi = {}                     # group index -> (parameter identities, ctypes pointer / count arrays)
        self._found = {}                     # device -> float32 [1]: this optimizer's own inf flag (step(grad_scaler=...))
        self._found2 = {}                    # device -> [float32 [2] flags used alternately, uint32 arrival counter, parity] (one-launch check + prologue)
        self._pending = []
        # GradScaler.step() looks for the `grad_scaler` keyword with inspect.signature(optimizer.step) on EVERY call (~35 us of
        # Python through torch's hook wrapper around step); a function that carries __signature__ answers from it
        step_fn = type(self).step
        if not hasattr(step_fn, "__signature__"):
            try:
                import inspect
                step_fn.__signature__ = inspect.signature(step_fn)
            except (TypeError, ValueError, AttributeError):
                pass

    def zero_grad(self, set_to_none=None):
        return super().zero_grad(self.set_grad_none if set_to_none is None else set_to_none)

    # The bias-correction step count and the skip counter live in small device tensors (the kernels read them; nothing is read
    # back).  They are part of the optimizer's state: a resume that dropped them would restart the bias correction.
    def state_dict(self):
        sd = super().state_dict()
        if self._sf is not None:
            sd["ngp_group_state"] = [(f.detach().clone(), i.detach().clone()) for f, i in zip(self._sf, self._si)]
        return sd

    def load_state_dict(self, state_dict):
        sd = dict(state_dict)
        gs = sd.pop("ngp_group_state", None)
        if gs is not None:
            # validated BEFORE anything is applied: a state of another optimizer layout must not be half-loaded
            if len(gs) != len(self.param_groups):
                raise ValueError("ngp_group_state has %d entries, this optimizer has %d parameter groups" % (len(gs), len(self.param_groups)))
            for f, i in gs:
                if f.numel() != 8 or i.numel() != 8:
                    raise ValueError("ngp_group_state entries must hold 8 elements each (got %d / %d)" % (f.numel(), i.numel()))
        super().load_state_dict(sd)
        if gs is not None:
            # torch's Optimizer.load_state_dict casts the per-parameter `state` entries to the parameters' device; these are ours, so
            # the same has to happen here: a checkpoint read with map_location='cpu' would otherwise hand HOST pointers to the Adam
            # kernels
            sf, si = [], []
            for (f, i), group in zip(gs, self.param_groups):
                dev = self._group_device(group)
                sf.append(f.detach().to(device=dev, dtype=torch.float32).clone().reshape(8))
                si.append(i.detach().to(device=dev, dtype=torch.int32).clone().reshape(8))
            self._sf, self._si = sf, si
        self._multi = {}

    @staticmethod
    def _group_device(group):
        devs = {p.device for p in group["params"]}
        if len(devs) != 1:
            raise NotImplementedError("compat FusedAdam: the parameters of one group must live on one device (got %s)" % sorted(map(str, devs)))
        dev = next(iter(devs))
        if dev.type != "cuda":
            raise NotImplementedError("compat FusedAdam: parameters must be CUDA tensors (libngp_hip has no CPU path)")
        return dev

    def _check(self, p, g):
        if (not p.is_cuda or p.dtype != torch.float32 or g.dtype != torch.float32 or not p.is_contiguous() or not g.is_contiguous()
                or p.numel() % 4 or p.data_ptr() % 16 or g.data_ptr() % 16):
            raise NotImplementedError("compat FusedAdam: parameters and gradients must be contiguous, 16-byte aligned fp32 "
                                      "CUDA tensors with a multiple of 4 elements (got {} {})".format(tuple(p.shape), p.dtype))

    @torch.no_grad()
    def step(self, closure=None, grad_scaler=None):
        """One optimisation step: per parameter group a one-thread prologue (skip decision, bias corrections) and ONE multi-tensor
        launch over every parameter that has a gradient (`ngp_adam_multi`; round 4: one launch per tensor).  Unlike apex / torch,
        the sweep leaves `p.grad` ZERO-FILLED (the unscale, the update and the clearing are one pass over p / g / m / v): a caller
        that accumulates gradients over several backward passes per step() is unaffected, one that inspects `p.grad` after step()
        sees zeros.

        grad_scaler (round 5): `torch.amp.GradScaler.step(optimizer)` passes itself to an optimizer that declares
        `_step_supports_amp_scaling` and whose step() takes this keyword -- apex's own FusedAdam has the same signature -- and then
        leaves the inf check to the optimizer.  It is done here as ONE read-only launch over the group's gradients
        (`ngp_check_finite_multi`) instead of torch's `_check_inf_per_device` (a pass that rewrites every gradient, behind ~40 us of
        Python per step); the flag is recorded where `GradScaler.update()` looks for it, the unscale happens inside the Adam sweep, and
        nothing is read back.  torch announces (FutureWarning, once) that it will stop passing the scaler one day; without the
        keyword the attributes `grad_scale` / `found_inf` it sets instead are honoured as before."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = self._L
        scale, found = getattr(self, "grad_scale", None), getattr(self, "found_inf", None)
        own_check = False
        if grad_scaler is not None and grad_scaler.is_enabled():
            from torch.amp.grad_scaler import OptState
            state = grad_scaler._per_optimizer_states[id(self)]
            if state["stage"] is OptState.READY:                   # nobody has unscaled / checked these gradients yet: do it here
                own_check = True
                scale = grad_scaler._get_scale_async()
            else:                                                  # scaler.unscale_(optimizer) ran: gradients are unscaled, flags recorded
                scale = None
                flags = list(state["found_inf_per_device"].values())
                if not flags:
                    raise RuntimeError("compat FusedAdam: GradScaler recorded no inf flag for this optimizer (unscale_ saw no gradients)")
                found = flags[0] if len(flags) == 1 else sum(f.to(flags[0].device) for f in flags)
        devices = {p.device for group in self.param_groups for p in group["params"] if p.grad is not None}
        if len(devices) > 1:
            # one skip decision per step needs one flag every group's prologue can read: parameters spread over devices would make the
            # prologues of the other devices dereference a foreign pointer
            raise NotImplementedError("compat FusedAdam: all parameters must live on one device (got %s)" % sorted(map(str, devices)))
        self._pending.clear()                  # (a previous step() that raised half-way must not be replayed with stale pointers)
        try:
            return self._step(L, scale, found, own_check, grad_scaler, loss)
        finally:
            self._pending.clear()

    def _step(self, L, scale, found, own_check, grad_scaler, loss):
        n_active = sum(1 for group in self.param_groups if any(p.grad is not None for p in group["params"]))
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            if self._sf is None:
                self._sf = [torch.zeros(8, device=self._group_device(g), dtype=torch.float32) for g in self.param_groups]
                self._si = [torch.zeros(8, device=self._group_device(g), dtype=torch.int32) for g in self.param_groups]
            sf, si = self._sf[gi], self._si[gi]
            b1, b2 = group["betas"]
            st = _stream()
            chunks = []
            for c0 in range(0, len(ps), _MULTI_MAX):
                chunk = ps[c0:c0 + _MULTI_MAX]
                key = (gi, c0)
                ident = tuple((id(p), p.data_ptr()) for p in chunk)
                ent = self._multi.get(key)
                if ent is None or ent[0] != ident:
                    k = len(chunk)
                    P, G, M, V = ((ctypes.c_void_p * k)() for _ in range(4))
                    N = (ctypes.c_longlong * k)()
                    for j, p in enumerate(chunk):
                        self._check(p, p.grad)
                        state_p = self.state[p]
                        if not state_p:
                            state_p["exp_avg"], state_p["exp_avg_sq"] = torch.zeros_like(p), torch.zeros_like(p)
                        P[j], M[j], V[j], N[j] = p.data_ptr(), state_p["exp_avg"].data_ptr(), state_p["exp_avg_sq"].data_ptr(), p.numel()
                    ent = self._multi[key] = (ident, P, G, M, V, N)
                _, P, G, M, V, N = ent
                for j, p in enumerate(chunk):                      # autograd hands out fresh gradient tensors every step
                    g = p.grad
                    if g.dtype != torch.float32 or not g.is_contiguous() or g.data_ptr() % 16:
                        self._check(p, g)
                    G[j] = g.data_ptr()
                chunks.append((len(chunk), P, G, M, V, N))
            merged = False
            if own_check and n_active == 1 and len(chunks) == 1 and scale is not None and scale.dtype == torch.float32:
                # the usual case (train.py:143-149: one group, six tensors): inf check and prologue in ONE launch, and no fill -- two flags
                # used alternately, each cleared by the launch of the step after the one GradScaler.update() read it in
                ent = self._found2.get(dev)
                if ent is None:
                    ent = self._found2[dev] = [torch.zeros(2, device=dev, dtype=torch.float32), torch.zeros(17 * 32, device=dev, dtype=torch.int32), 0]
                pair, done, par = ent
                found, nxt = pair[par:par + 1], pair[1 - par:2 - par]
                ent[2] = 1 - par
                grad_scaler._per_optimizer_states[id(self)]["found_inf_per_device"][dev] = found
                k, P, G, M, V, N = chunks[0]
                _lib.check(L.ngp_adam_amp_check_prologue(k, G, N, _ptr(found), _ptr(nxt), _ptr(done), _ptr(sf), _ptr(si), _ptr(scale),
                                                         float(group["lr"]), float(b1), float(b2), st), "ngp_adam_amp_check_prologue")
                merged = True
            elif own_check:
                if gi == 0 or found is None or found.device != dev:
                    found = self._found.get(dev)
                    if found is None:
                        found = self._found[dev] = torch.zeros(1, device=dev, dtype=torch.float32)
                    else:
                        found.zero_()
                    grad_scaler._per_optimizer_states[id(self)]["found_inf_per_device"][dev] = found
                for k, P, G, M, V, N in chunks:
                    _lib.check(L.ngp_check_finite_multi(k, G, N, _ptr(found), st), "ngp_check_finite_multi")
            if scale is not None and (scale.dtype != torch.float32 or found is None or found.dtype != torch.float32):
                raise TypeError("grad_scale / found_inf must be float32 device tensors (torch.cuda.amp.GradScaler's are)")
            self._pending.append((gi, group, sf, si, b1, b2, chunks, st, ps, scale, found, merged))
        # every group's flag is complete before any group is updated: one overflow skips the whole step (GradScaler's semantics)
        for gi, group, sf, si, b1, b2, chunks, st, ps, scale, found, merged in self._pending:
            if not merged:
                _lib.check(L.ngp_adam_amp_prologue(_ptr(sf), _ptr(si), _ptr(scale), _ptr(found), float(group["lr"]), float(b1), float(b2),
                                                   st), "ngp_adam_amp_prologue")
            for k, P, G, M, V, N in chunks:
                _lib.check(L.ngp_adam_multi(k, P, G, M, V, N, _ptr(sf), _ptr(si), float(b1), float(b2), float(group["eps"]), st),
                           "ngp_adam_multi")
            _touched(*ps, *[p.grad for p in ps])            # written through raw pointers: version-keyed caches (the encoders' 16-bit table copies) must see it
        return loss


_MULTI_MAX = 16                      # NGP_ADAM_MULTI_MAX (include/ngp_hip.h)
