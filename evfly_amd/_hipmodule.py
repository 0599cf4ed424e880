"""Handle management shared by the nn.Module shells.

The shells own ordinary torch parameter containers (so `state_dict()` keys, `.to()`,
`.float()`, `load_state_dict()` and `.eval()` behave exactly like the reference's
modules, evfly_ros/run.py:120-171) but never run them: `forward` hands the tensors to
the C ABI (`evfly_model_*`, include/evfly_hip.h) which repacks them for the HIP
kernels. The handle is rebuilt lazily whenever the parameters may have changed.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib


class HipHandle:
    """RAII wrapper of an `evfly_model*`."""

    def __init__(self, cfg, state_dict):
        L = _lib.lib()
        self._L = L
        self.h = C.c_void_p()
        _lib.check(L.evfly_model_create(C.byref(cfg), C.byref(self.h)))
        for k, v in state_dict.items():
            t = v.detach().to("cpu", torch.float32).contiguous()
            shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
            _lib.check(L.evfly_model_load_tensor(self.h, k.encode(), t.data_ptr(), shape, t.dim()))
        _lib.check(L.evfly_model_finalize(self.h))

    def __del__(self):
        try:
            if self.h:
                self._L.evfly_model_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def tap(self, name, max_elems=1 << 26):
        """Copy a named intermediate of the last forward to host (parity tests). "e1".."e4" are complete only with
        EVFLY_FULL_ENCODER_OUTPUTS=1 in the environment (include/evfly_hip.h, evfly_model_tap)."""
        buf = torch.empty(max_elems, dtype=torch.float32)
        shape = (C.c_int64 * 4)()
        n = _lib.check(self._L.evfly_model_tap(self.h, name.encode(), buf.data_ptr(), max_elems, shape,
                                               _lib.cur_stream()))
        dims = [int(s) for s in shape if s > 0]
        return buf[:n].reshape(dims).clone()

    def profile(self):
        L = self._L
        out = []
        name = C.create_string_buffer(128)
        ms, fl, by, ln = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        for i in range(L.evfly_model_profile_count(self.h)):
            L.evfly_model_profile_get(self.h, i, name, 128, C.byref(ms), C.byref(fl), C.byref(by), C.byref(ln))
            ex = C.c_double()
            L.evfly_model_profile_exec_flops(self.h, i, C.byref(ex))
            us = C.c_double()
            L.evfly_model_profile_useful_flops(self.h, i, C.byref(us))
            out.append(dict(name=name.value.decode(), ms=ms.value, flops=fl.value, bytes=by.value,
                            launches=ln.value, exec_flops=ex.value, useful_flops=us.value))
        return out


class HipModule(nn.Module):
    """Base of the shells: lazily (re)builds the native handle from the current parameters."""

    compute_dtype = 0  # _lib EVFLY_DTYPE_F32; set to 1 for bf16 MFMA operands

    def __init__(self):
        super().__init__()
        self.__dict__["_hip"] = None
        self.__dict__["_hip_built_at"] = None
        self.__dict__["_epoch"] = 0
        self.__dict__["_slot_cache"] = None
        self.__dict__["_frozen"] = False

    # ---- invalidation: the native handle holds a packed COPY of the weights, so it is stale whenever any parameter
    # or buffer of this module tree changed. Overriding entry points (load_state_dict, _apply) is not enough:
    # `parent.load_state_dict(ckpt)` recurses through `_load_from_state_dict` and never calls the children's
    # `load_state_dict`, and `optimizer.step()` / `p.copy_()` / `p.data = ...` touch no module method at all. The handle
    # is therefore keyed by a fingerprint of every tensor: (storage pointer, autograd version counter) -- in-place
    # writes through the tensor bump the version, re-assignment / `.to()` changes the pointer -- plus the compute dtype
    # and an explicit epoch for the writes torch does not record there (`refresh_weights()`).
    # NOT seen by the fingerprint: in-place writes made through `.data` (`p.data.copy_(w)`, `p.data.mul_(a)`, the usual EMA
    # update): `.data` is a detached alias with its own version counter, so `p._version` and the pointer both stay put.
    # Call `refresh_weights()` after such writes.
    def _slots(self):
        """(dict, key) of every parameter / buffer slot of the tree, collected once: the per-forward check then costs one
        dict lookup per tensor instead of a walk over `modules()` (0.35 ms for the composite's 149 tensors, in front of
        the first launch of a 1.9 ms single-frame forward). Looking the tensor up through its slot also catches a
        re-assigned Parameter object. The list is dropped by `_invalidate` (load_state_dict, _apply, register_* / add_module /
        module or Parameter assignment on a shell), so slots that were None at the first forward or sub-modules registered later
        on a shell enter it; a child registered on a plain nn.Module INSIDE a shell after the first forward needs
        `refresh_weights()`."""
        sl = self.__dict__.get("_slot_cache")
        if sl is None:
            sl, seen = [], set()
            for mod in self.modules():
                for d in (mod._parameters, mod._buffers):
                    for k, t in d.items():
                        if t is not None and id(t) not in seen:
                            seen.add(id(t)); sl.append((d, k))
            self.__dict__["_slot_cache"] = sl
        return sl

    def _fingerprint(self):
        fp = [self.compute_dtype, self.__dict__["_epoch"]]
        for d, k in self._slots():
            t = d[k]
            fp.append(t.data_ptr()); fp.append(t._version)
        return hash(tuple(fp))

    def refresh_weights(self):
        """Force a rebuild of the packed native weights at the next forward. Needed after writes the fingerprint cannot
        see: in-place writes through `.data` (`p.data.copy_()`, EMA updates) and writes through a raw pointer."""
        self.__dict__["_epoch"] += 1
        self.__dict__["_slot_cache"] = None
        self.__dict__["_frozen"] = False

    def freeze(self):
        """Deployment fast path: build the handle now and stop checking the parameters on every forward (a 15 Hz node
        never touches them, evfly_ros/run.py:171). `refresh_weights()` or `set_compute_dtype()` thaw it."""
        self.__dict__["_frozen"] = False
        self.hip()
        self.__dict__["_frozen"] = True
        return self

    # ---- every structural or bulk weight change that goes through a module method thaws a frozen handle and drops the cached
    # slot list (cheap, not on the forward path): .to() / .float() / .cuda() (_apply), load_state_dict on this module or on a
    # parent (_load_from_state_dict is what the recursion calls), registering a parameter / buffer / sub-module later on.
    def _invalidate(self):
        d = self.__dict__
        if "_slot_cache" in d:
            d["_slot_cache"] = None
            d["_frozen"] = False
        for m in self.__dict__.get("_modules", {}).values():        # shells nested inside a shell (the composite)
            if isinstance(m, HipModule):
                m._invalidate()

    def _invalidate_tree(self):
        """thaw this shell and every HipModule above / below it that shares these tensors"""
        self._invalidate()
        for owner in self.__dict__.get("_hip_owners", ()):
            o = owner()
            if o is not None:
                o._invalidate()

    def _apply(self, fn, *a, **kw):
        self._invalidate_tree()
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self._invalidate_tree()
        return super().load_state_dict(*a, **kw)

    def _load_from_state_dict(self, *a, **kw):
        self._invalidate_tree()
        return super()._load_from_state_dict(*a, **kw)

    def register_parameter(self, name, param):
        self._invalidate_tree()
        return super().register_parameter(name, param)

    def register_buffer(self, name, tensor, persistent=True):
        self._invalidate_tree()
        return super().register_buffer(name, tensor, persistent=persistent)

    def add_module(self, name, module):
        self._invalidate_tree()
        if isinstance(module, HipModule):
            import weakref
            module.__dict__.setdefault("_hip_owners", []).append(weakref.ref(self))
        return super().add_module(name, module)

    def __setattr__(self, name, value):
        if isinstance(value, (nn.Module, nn.Parameter)):
            self._invalidate_tree()
            if isinstance(value, HipModule):
                import weakref
                value.__dict__.setdefault("_hip_owners", []).append(weakref.ref(self))
        return super().__setattr__(name, value)

    def set_compute_dtype(self, name):
        self.compute_dtype = {"f32": 0, "fp32": 0, "bf16": 1, "bf16x3": 2}[name]
        self.__dict__["_frozen"] = False
        return self

    # ---- subclasses provide the config + key prefixing
    def _hip_config(self):
        raise NotImplementedError

    def _hip_state_dict(self):
        return self.state_dict()

    def hip(self):
        h = self.__dict__["_hip"]
        if h is not None and self.__dict__.get("_frozen"):
            return h
        fp = self._fingerprint()
        if h is None or self.__dict__["_hip_built_at"] != fp:
            h = HipHandle(self._hip_config(), self._hip_state_dict())
            self.__dict__["_hip"] = h
            self.__dict__["_hip_built_at"] = fp
        return h


_WARNED_NO_GRAD = set()


def _inference_only(mod, what, training_differs):
    """The stand-alone native forwards are INFERENCE-ONLY: weights are read detached (no autograd graph comes back), BatchNorm uses
    its running statistics and Dropout is the identity whatever `mod.training` says. Where that changes the numbers the reference
    would produce (a module in training mode that holds BatchNorm / an active Dropout) this raises instead of returning them
    silently; a call that merely expects gradients is warned about once per class."""
    _lib.lib()                                                   # (no GPU / no library: that error first)
    if mod.training and training_differs:
        raise NotImplementedError(f"[{what}] the native forward is inference-only (BatchNorm running statistics, Dropout off): "
                                  f"call .eval() first; training-mode semantics (batch statistics, dropout masks, running-stat "
                                  f"updates) are not implemented")
    if torch.is_grad_enabled() and what not in _WARNED_NO_GRAD and any(p.requires_grad for p in mod.parameters()):
        _WARNED_NO_GRAD.add(what)
        import warnings
        warnings.warn(f"[{what}] the native forward is inference-only: the outputs carry no autograd graph (gradients of these "
                      f"parameters will be None); wrap the call in torch.no_grad() to silence this", stacklevel=3)


def to_gpu(t, dtype=torch.float32):
    """Input staging (evfly_ros/run.py:247 `.to(self.device).float()`): returns a contiguous CUDA
    tensor; host tensors are copied to the current device."""
    _lib.lib()
    if not t.is_cuda:
        t = t.to("cuda")
    return t.to(dtype).contiguous()
