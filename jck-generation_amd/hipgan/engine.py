"""Python face of the native step engine (jck_engine_* in include/jckgan.h).

The engine owns no memory: this wrapper allocates the flat fp32 arenas (parameters, gradients, Adam
moments, BatchNorm running statistics) and one workspace as torch tensors on the device and binds
them.  `named_views()` exposes every tensor under the reference's state-dict key
(model/DCGAN.py:10-27,42-59) as a zero-copy view, so nn.Module parameters can live in the arenas.
"""
import ctypes as C
import os

import torch

from ._lib import PREC_BF16, PREC_F32, JckError, StepInputs, cur_stream, lib, load_library

(PHASE_D_LOSS, PHASE_D_GP, PHASE_D_STEP, PHASE_G_LOSS, PHASE_G_STEP, PHASE_D_REAL, PHASE_D_FAKE, PHASE_D_REAL_FWD, PHASE_D_LOSS_A,
 PHASE_D_LOSS_B) = range(10)
PHASE_LAZY_JOIN = 0x100
PHASE_GP_ONLY = 10                 # include/jckgan.h: the gradient penalty alone (module path)
PHASE_NO_RESIDENT = 0x200          # include/jckgan.h: no grid-barrier launch in this phase call (a collective may be holding CUs)
_PREC = {"bf16": PREC_BF16, "f32": PREC_F32, PREC_BF16: PREC_BF16, PREC_F32: PREC_F32}
SCALAR_NAMES = ("loss_d", "loss_g", "d_x", "d_gz1", "d_gz2", "gp", "loss_real", "loss_fake")


def _layout(handle, net):
    """[(state-dict key, kind, arena offset, numel, shape)] of a created engine's network (its own image size)."""
    dll = load_library()
    out = []
    for i in range(dll.jck_engine_num_tensors_of(handle, net)):
        name = C.create_string_buffer(64)
        kind, off, numel = C.c_int(), C.c_longlong(), C.c_longlong()
        shape = (C.c_int * 4)()
        rc = dll.jck_engine_tensor_info_of(handle, net, i, name, 64, C.byref(kind), C.byref(off), C.byref(numel), shape)
        if rc != 0:
            raise JckError(dll.jck_last_error().decode())
        shp = list(shape)
        nm = name.value.decode()
        if kind.value == 0 and nm.startswith("conv"):
            shp = shp
        elif kind.value == 0 and nm.endswith(".weight") and (nm.startswith("linear") or nm.startswith("label_embedding")):
            shp = shp[:2]
        else:
            shp = shp[:1]
        out.append((name.value.decode(), kind.value, off.value, numel.value, shp))
    return out


# Engines are destroyed by the garbage collector, which can run at any allocation - also in the middle of a stream capture of
# ANOTHER engine.  Destroying streams / events / graphs there (hipStreamSynchronize, hipStreamDestroy, hipGraphExecDestroy) is
# an "unsafe call" inside a capture: it invalidates the graph being captured and the next hipGraphLaunch of it crashes.  So
# destruction is deferred while a capture is open.
_CAPTURES_OPEN = 0
_DEFERRED_DESTROY = []


def _destroy_native(graphs, handle):
    dll = load_library()
    for ge in graphs:
        dll.jck_graph_destroy(ge)
    if handle:
        dll.jck_engine_destroy(handle)


def _flush_deferred():
    while _DEFERRED_DESTROY and _CAPTURES_OPEN == 0:
        _destroy_native(*_DEFERRED_DESTROY.pop())


class _GraphUnavailable(JckError):
    """Capture of a step segment failed before anything of the step ran: the engine falls back to eager launches."""


class DeviceBatch:
    """A training batch as indices into a uint8 image dataset that lives in HBM ([N,3,32,32], the CIFAR pickle layout).
    The step gathers and transforms it on the device (jck_img_prep_u8: the reference's Resize(64) / ToTensor /
    Normalize(0.5, 0.5), preprocess/dcgan_data_preprocessor.py:38-43, bit-exact) - no per-step host->device image copy."""

    def __init__(self, data_u8, idx):
        if data_u8.dtype != torch.uint8 or not data_u8.is_cuda or not data_u8.is_contiguous():
            raise JckError("DeviceBatch: dataset must be a contiguous uint8 CUDA tensor")
        self.data = data_u8
        self.idx = idx.to(data_u8.device, torch.int64).contiguous()

    def size(self, dim=0):
        return (self.idx.numel(), 3, 64, 64)[dim]

    def materialize(self):
        """The transformed batch as fp32 NCHW [B,3,64,64] (what the reference's DataLoader would have yielded)."""
        b = self.idx.numel()
        out = torch.empty(b, 3, 64, 64, dtype=torch.float32, device=self.data.device)
        lib.jck_img_prep_u8(PREC_F32, self.data, self.idx, None, 1.0, 0.0, None, out, b, 32, 32, cur_stream())
        return out


class DcganEngine:
    """One DCGAN training state resident in HBM + the native step schedule."""

    family = 0

    def __init__(self, batch, prec="bf16", device="cuda:0", share=None, image_size=64):
        """share: another DcganEngine whose arenas (weights, gradients, Adam moments, BN statistics) this one binds
        too - used for the ragged last batch of an epoch, which needs its own workspace geometry but the same state.
        image_size: 64 = the reference's nets; 128 = one more stride-2 stage (DCGAN only, BASELINE.json configs[4])."""
        if not torch.cuda.is_available():
            raise JckError("DcganEngine needs a GPU: the HIP path has no CPU fallback")
        self.device = torch.device(device) if share is None else share.device
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        torch.cuda.set_device(self.device)
        self.prec = _PREC[prec] if share is None else share.prec
        self.batch = batch
        self.size = image_size if share is None else share.size
        self._shared = share._shared if share is not None else {"t": 0, "version": 0, "last_step": 0}
        self._packed_version = -1
        # hipGraph replay of the step (JCK_GRAPH=1 enables): one captured graph per (segment, step parity, input kind).
        # Default: off.  A captured step is linear (see jck_engine_phase), so it gives up the second stream that runs the weight
        # gradients beside the dgrad chain (+8 % DCGAN, +1.4 % CGAN), and the host is not the bottleneck: 1.15 ms of enqueue for
        # DCGAN's 1.8 ms step, 1.9 ms for CGAN's 2.9 ms (round 3, tools/ab.sh: CGAN eager 2.886 ms, replayed 2.97 ms - in round 2
        # the eager CGAN step was enqueue-bound at 3.4 ms and the graph was its default).  Replay remains the answer when the host
        # is busy or slow: 0.10-0.13 ms of host time per step.
        self.graphs = os.environ.get("JCK_GRAPH", "0") != "0"
        # data parallel: D's all-reduce in two pieces under D's own backward + G's under the next batch's D(real) forward;
        # False = one all-reduce per network, waited for before its Adam (hipgan.dist.ReplicaGuard falls back to it)
        self.lazy_join = os.environ.get("JCK_LAZY_JOIN", "1") != "0"
        self.ddp_lazy_tail = os.environ.get("JCK_DDP_LAZY_TAIL", "1") != "0"
        self._ar_stream = None
        self.ddp_overlap = os.environ.get("JCK_DDP_SPLIT", "1") == "1"
        # steps without a noise dict draw z / alpha with torch and the two instance-noise tensors INSIDE the image kernels
        # (Philox, jck_engine_set_noise_seed); JCKGAN_FAST_NOISE=0 draws them with torch.randn as round 1 did
        self.fast_noise = os.environ.get("JCKGAN_FAST_NOISE", "1") != "0"
        self._graph_cache, self._sbuf, self._st, self._eager_steps = {}, None, None, 0
        h = C.c_void_p()
        dll = load_library()
        if dll.jck_engine_create_sized(C.byref(h), self.family, self.prec, batch, self.size) != 0:
            raise JckError(dll.jck_last_error().decode())
        self._h = h
        f32 = dict(dtype=torch.float32, device=self.device)
        self.arenas = {} if share is None else share.arenas
        for net, tag in ((0, "g"), (1, "d")):
            if share is not None:
                break
            n, nb = dll.jck_engine_arena_numel_of(h, net, 0), dll.jck_engine_arena_numel_of(h, net, 1)
            for what in ("params", "grads", "m", "v"):
                self.arenas[f"{tag}_{what}"] = torch.zeros(n, **f32)
            self.arenas[f"{tag}_bn"] = torch.zeros(nb, **f32)
            self.arenas[f"{tag}_nbt"] = torch.zeros(8, dtype=torch.int64, device=self.device)
        self.ws_bytes = dll.jck_engine_workspace_bytes(h)
        self.workspace = torch.zeros(self.ws_bytes, dtype=torch.uint8, device=self.device)
        a = self.arenas
        lib.jck_engine_bind(h, self.workspace, self.ws_bytes, a["g_params"], a["g_grads"], a["g_m"], a["g_v"], a["g_bn"],
                            a["g_nbt"], a["d_params"], a["d_grads"], a["d_m"], a["d_v"], a["d_bn"], a["d_nbt"])
        self.layout = {"g": _layout(h, 0), "d": _layout(h, 1)}
        # the step's own draws (Philox: z, alpha, instance noise, dropout masks) follow torch's seed unless the caller sets one
        self.set_noise_seed(int(torch.initial_seed()) + 0x6a636b67)
        # BN running_var starts at 1 (nn.BatchNorm2d)
        for tag in ("g", "d"):
            for name, kind, off, numel, shp in self.layout[tag]:
                if kind == 2 and share is None:
                    a[f"{tag}_bn"][off:off + numel].fill_(1.0)

    # optimiser step count and weight version are shared by engines bound to the same arenas
    @property
    def t(self):
        return self._shared["t"]

    @t.setter
    def t(self, v):
        self._shared["t"] = v

    def __del__(self):
        try:
            graphs, handle = list(getattr(self, "_graph_cache", {}).values()), getattr(self, "_h", None)
            self._graph_cache, self._h = {}, None
            if _CAPTURES_OPEN > 0:
                _DEFERRED_DESTROY.append((graphs, handle))
            else:
                _destroy_native(graphs, handle)
        except Exception:
            pass

    # ---- state ------------------------------------------------------------------------------------------
    def named_views(self, tag, what="params"):
        """{state-dict key: view}. what: 'params' (incl. BN buffers), 'grads', 'm', 'v'.  (Call join() first when a step
        may still be in flight and the views are about to be read.)"""
        out = {}
        bn_i = 0
        for name, kind, off, numel, shp in self.layout[tag]:
            if kind == 0:
                out[name] = self.arenas[f"{tag}_{what}"][off:off + numel].view(shp)
            elif what == "params":
                out[name] = self.arenas[f"{tag}_bn"][off:off + numel].view(shp)
                if kind == 2:
                    out[name.replace("running_var", "num_batches_tracked")] = self.arenas[f"{tag}_nbt"][bn_i]
                    bn_i += 1
        return out

    def load_state(self, g_state, d_state):
        """Copies reference-keyed state dicts (CPU or device tensors) into the arenas and re-derives the bf16 operands."""
        self.join()
        for tag, sd in (("g", g_state), ("d", d_state)):
            views = self.named_views(tag)
            for k, v in sd.items():
                if k not in views:
                    raise JckError(f"unexpected key {k}")
                views[k].copy_(v.detach().to(self.device).view(views[k].shape))
        self.mark_weights_changed()
        self.repack()

    def state_dicts(self):
        self.join()
        order = lambda tag: {k: v.detach().cpu().clone() for k, v in self._ordered(tag)}
        return order("g"), order("d")

    def _ordered(self, tag):
        """reference state_dict order: per layer weight, [bias, running_mean, running_var, num_batches_tracked]."""
        v = self.named_views(tag)
        keys = [k for k in ("label_embedding.weight", "label_embedding.bias") if k in v]
        for i in range(1, 8):
            if f"conv{i}.weight" in v:
                keys.append(f"conv{i}.weight")
            if f"norm{i}.weight" in v:
                keys += [f"norm{i}.weight", f"norm{i}.bias", f"norm{i}.running_mean", f"norm{i}.running_var",
                         f"norm{i}.num_batches_tracked"]
        keys += [k for k in ("linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias") if k in v]
        return [(k, v[k]) for k in keys]

    def repack(self):
        """Re-derive this engine's GEMM operand copies from the fp32 parameters (after any out-of-engine write)."""
        lib.jck_engine_repack(self._h, 0, cur_stream())
        lib.jck_engine_repack(self._h, 1, cur_stream())
        self._packed_version = self._shared["version"]

    def mark_weights_changed(self):
        self._shared["version"] += 1
        if getattr(self, "_prefetched_real", None) is not None:
            # the next step's D(real) forward is already in flight with the OLD weights (step_async(next_real=...)): the current
            # stream waits for it before anything overwrites or repacks them, and the next step computes it afresh
            lib.jck_engine_drop_prefetch(self._h, torch.cuda.current_stream().cuda_stream)
            self._prefetched_real = None

    def adopt_modules(self, model_g, model_d):
        """Moves the parameters / buffers of reference-shaped nn.Modules INTO the arenas (zero copy afterwards):
        `module.state_dict()`, checkpoints and `.grad` then always show the live training state."""
        for tag, mod in (("g", model_g), ("d", model_d)):
            views, grads = self.named_views(tag), self.named_views(tag, "grads")
            with torch.no_grad():
                for name, p in mod.named_parameters():
                    views[name].copy_(p.detach().to(self.device))
                    p.data = views[name]
                    p.grad = grads[name]
                for name, b in list(mod.named_buffers()):
                    views[name].copy_(b.detach().to(self.device))
                    owner = mod
                    *path, leaf = name.split(".")
                    for part in path:
                        owner = getattr(owner, part)
                    owner._buffers[leaf] = views[name]
        self.mark_weights_changed()
        self.repack()

    # ---- the step ---------------------------------------------------------------------------------------
    def _inputs(self, real, noise, lr, grad_scale):
        B, S = self.batch, self.size
        si = StepInputs()
        keep = []
        if isinstance(real, DeviceBatch):           # indices into a uint8 dataset resident in HBM: the step transforms them itself
            if real.size(0) != B or tuple(real.data.shape[1:]) != (3, 32, 32):
                raise JckError(f"DeviceBatch must index {B} images of a uint8 [N,3,32,32] dataset")
            keep += [real.data, real.idx]
            si.real_u8, si.real_idx = real.data.data_ptr(), real.idx.data_ptr()
            real = None
        elif real is not None and (real.shape != (B, 3, S, S) or real.dtype != torch.float32):
            raise JckError(f"real must be float32 [{B},3,{S},{S}], got {tuple(real.shape)} {real.dtype}")

        def ptr(t, shape):
            if t is None:
                return None
            t = t.to(self.device, torch.float32).contiguous()
            if t.numel() != shape:
                raise JckError(f"noise tensor has {t.numel()} elements, expected {shape}")
            keep.append(t)
            return t.data_ptr()
        si.real_nchw = ptr(real, B * 3 * S * S)
        si.noise_real = ptr(noise.get("n1"), B * 3 * S * S)
        si.z = ptr(noise.get("z"), B * 100)
        si.noise_fake = ptr(noise.get("n2"), B * 3 * S * S)
        si.alpha = ptr(noise.get("alpha"), B)
        si.lr, si.grad_scale, si.step = lr, grad_scale, self.t + 1
        if self.family == 1:
            labels = noise.get("labels")
            if labels is None or labels.shape != (B, 100) or labels.dtype != torch.int64:
                raise JckError("CGAN step needs labels: int64 one-hot [B,100]")
            lab = labels.to(self.device).contiguous()
            keep.append(lab)
            si.labels = lab.data_ptr()
            ms = [noise.get(f"m{i + 1}") for i in range(4)]
            if all(m is None for m in ms):
                return si, keep                     # the engine draws its own masks (and z / alpha when those are None too)
            if any(m is None for m in ms):
                raise JckError("CGAN step needs all four dropout keep-masks m1..m4 [B,256] (or none: drawn by the engine)")
            # the engine runs the head of the real | fake | penalty groups as one 3B-row pass when their masks lie back to
            # back: hand them over as one [4, B, 256] tensor (draw_noise already makes them that way)
            adjacent = all(torch.is_tensor(m) and m.is_cuda and m.dtype == torch.float32 and m.is_contiguous() and m.numel() == B * 256
                           for m in ms) and all(ms[i + 1].data_ptr() == ms[i].data_ptr() + B * 256 * 4 for i in range(3))
            if not adjacent:
                mm = torch.stack([m.to(self.device, torch.float32).reshape(B, 256) for m in ms])
                keep.append(mm)
                ms = [mm[i] for i in range(4)]
            for i in range(4):
                si.drop_mask[i] = ptr(ms[i], B * 256)
        return si, keep

    def set_noise_seed(self, seed):
        """Seed of the in-kernel instance noise (data-parallel ranks: base seed + rank)."""
        lib.jck_engine_set_noise_seed(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF)

    def draw_noise(self, generator=None, labels=None, out=None, fast=False):
        """Device-side draws in the reference's order (train/dcgan_trainer.py:160,168,171,111).  out: the engine's
        fixed-address input buffers (graph replay) - filled in place with the same draws, no extra copy.  fast: only z and
        alpha (and the dropout masks); the two [B,3,S,S] instance-noise tensors are then drawn inside the step's kernels."""
        B, dev, S = self.batch, self.device, self.size
        if fast:
            # perf mode: NOTHING is drawn here - z, alpha and CGAN's dropout masks come out of the engine's per-step launch
            # (jck_engine_set_step: Philox keyed by set_noise_seed and the step), the two instance-noise tensors out of the
            # image kernels; `generator` plays no part
            nz = {"n1": None, "z": None, "n2": None, "alpha": None}
            if self.family == 1:
                nz["labels"] = labels
            return nz
        # one normal draw for n1 | z | n2 (three launches -> one; the order inside the buffer is the reference's)
        ni, nzz = B * 3 * S * S, B * 100
        if out is None:
            buf = torch.randn(2 * ni + nzz, device=dev, generator=generator)
            alpha = torch.rand(B, 1, 1, 1, device=dev, generator=generator)
        else:
            buf = torch.randn(2 * ni + nzz, generator=generator, out=out["nbuf"])
            alpha = torch.rand(B, 1, 1, 1, generator=generator, out=out["alpha"])
        nz = {"n1": buf[:ni].view(B, 3, S, S),
              "z": buf[ni:ni + nzz].view(B, 100, 1, 1),
              "n2": buf[ni + nzz:].view(B, 3, S, S),
              "alpha": alpha}
        if self.family == 1:
            nz["labels"] = labels
            self._draw_masks(nz, generator, out)
        return nz

    def _draw_masks(self, nz, generator, out):
        """nn.Dropout(0.25) keep masks of the four D passes (model/CGAN.py:105) from ONE uniform draw [4, B, 256] (three launches
        instead of twelve: the step is a serial chain, every ~5 us launch counts)."""
        B, dev = self.batch, self.device
        if out is None:
            m = (torch.rand(4, B, 256, device=dev, generator=generator) >= 0.25).float()
        else:
            u = torch.rand(4, B, 256, generator=generator, out=out["u"])
            m = out["m"].copy_(u >= 0.25)
        for i in range(4):
            nz[f"m{i + 1}"] = m[i]

    def join(self):
        """Makes the current stream wait for engine work still in flight on the engine's own stream - the graph-replay stream
        (call before reading weights, scalars, ...)."""
        cur = torch.cuda.current_stream()
        for key in ("e_stream",):
            s_ = self._shared.get(key)
            if s_ is not None and s_ != cur:
                cur.wait_stream(s_)

    # ---- hipGraph replay -----------------------------------------------------------------------------------
    # A step is ~125 (DCGAN) / ~280 (CGAN) kernel launches; replayed from a captured graph the host issues ONE call per
    # step segment.  A captured graph bakes every kernel argument, so: the step inputs live in fixed-address buffers that are
    # refilled in place, the Adam bias corrections go through device memory (jck_engine_set_step), and there is one graph
    # per step parity (the scalar and BatchNorm-record buffers alternate).  All of it runs on the engine's own stream.
    def _e_stream(self):
        if self._shared.get("e_stream") is None:
            self._shared["e_stream"] = torch.cuda.Stream(device=self.device)
        return self._shared["e_stream"]

    def _static(self):
        if self._sbuf is None:
            B, dev, S = self.batch, self.device, self.size
            f32 = dict(dtype=torch.float32, device=dev)
            sb = {"nbuf": torch.empty(2 * B * 3 * S * S + B * 100, **f32), "alpha": torch.empty(B, 1, 1, 1, **f32),
                  "z": torch.empty(B, 100, 1, 1, **f32),
                  "real": torch.empty(B, 3, S, S, **f32), "idx": torch.empty(B, dtype=torch.int64, device=dev)}
            if self.family == 1:
                sb["labels"] = torch.empty(B, 100, dtype=torch.int64, device=dev)
                sb["u"] = torch.empty(4, B, 256, **f32)
                sb["m"] = torch.empty(4, B, 256, **f32)
                for i in range(4):
                    sb[f"m{i + 1}"] = sb["m"][i]              # one address per mask whether it is drawn or handed in
            self._sbuf = sb
        return self._sbuf

    def _fill_static(self, real, noise, generator, labels):
        """Puts this step's inputs into the fixed-address buffers (on the current = engine stream); -> (real, noise) that
        point into them."""
        sb, B, S = self._static(), self.batch, self.size
        ni, nzz = B * 3 * S * S, B * 100
        if noise is None:
            if self.family == 1:
                if labels is None:
                    raise JckError("CGAN step without a noise dict needs labels=")
                sb["labels"].copy_(labels.to(torch.int64).view(B, 100), non_blocking=True)
            nz = self.draw_noise(generator, labels=sb.get("labels"), out=sb, fast=self.fast_noise)
        else:
            buf = sb["nbuf"]
            up = noise.get("n1") is not None or noise.get("n2") is not None      # uploaded instance noise (else: drawn in-kernel)
            if up:
                buf[:ni].view(B, 3, S, S).copy_(noise["n1"], non_blocking=True)
                buf[ni + nzz:].view(B, 3, S, S).copy_(noise["n2"], non_blocking=True)
            zin, ain = noise.get("z"), noise.get("alpha")
            if zin is not None:
                buf[ni:ni + nzz].view(B, 100, 1, 1).copy_(zin.view(B, 100, 1, 1), non_blocking=True)
            if ain is not None:
                sb["alpha"].copy_(ain.view(B, 1, 1, 1), non_blocking=True)
            nz = {"n1": buf[:ni].view(B, 3, S, S) if up else None, "z": buf[ni:ni + nzz].view(B, 100, 1, 1) if zin is not None else None,
                  "n2": buf[ni + nzz:].view(B, 3, S, S) if up else None, "alpha": sb["alpha"] if ain is not None else None}
            if self.family == 1:
                lab = noise.get("labels")
                if lab is None or lab.shape != (B, 100) or lab.dtype != torch.int64:
                    raise JckError("CGAN step needs labels: int64 one-hot [B,100]")
                nz["labels"] = sb["labels"].copy_(lab, non_blocking=True)
                for i in range(4):
                    m = noise.get(f"m{i + 1}")
                    if m is not None:
                        nz[f"m{i + 1}"] = sb[f"m{i + 1}"].copy_(m.view(B, 256), non_blocking=True)
        if isinstance(real, DeviceBatch):
            if real.size(0) != B:
                raise JckError(f"DeviceBatch must index {B} images")
            real = DeviceBatch(real.data, sb["idx"].copy_(real.idx, non_blocking=True))
        else:
            if real.shape != (B, 3, S, S) or real.dtype != torch.float32:
                raise JckError(f"real must be float32 [{B},3,{S},{S}], got {tuple(real.shape)} {real.dtype}")
            real = sb["real"].copy_(real, non_blocking=True)
        return real, nz

    def _step_graph(self, real, noise, lr, reduce_d, reduce_g, grad_scale, generator, labels):
        main, est = torch.cuda.current_stream(), self._e_stream()
        est.wait_stream(main)                                       # inputs produced on the caller's stream
        h, st = self._h, est.cuda_stream
        with torch.cuda.stream(est):
            real_s, nz = self._fill_static(real, noise, generator, labels)
            self._fallback_inputs = (real_s, nz)        # an eager retry must not draw a second time
            # The caller's tensors were read by copies on THIS stream: the caller's stream waits for those copies, so memory
            # it frees and reuses afterwards cannot be overwritten under them.  (Tensor.record_stream would do the same, but
            # the caching allocator then records an event on this stream whenever such a tensor is freed - also in the
            # middle of a capture, where that event becomes a node of the graph and is gone by the time the graph replays.)
            copied = torch.cuda.Event()
            copied.record(est)
            main.wait_event(copied)
            si, keep = self._inputs(real_s, nz, lr, grad_scale)
            step = self.t + 1
            lib.jck_engine_set_step(h, step, lr, st)
            # a captured segment must join every side stream it forks: the penalty pass started in PHASE_D_LOSS (per-pass
            # schedule) is joined by PHASE_D_GP, so the two always share a segment
            if reduce_d or reduce_g:
                segs = ([[PHASE_D_LOSS, PHASE_D_GP]], [[PHASE_D_STEP, PHASE_G_LOSS]], [[PHASE_G_STEP]])
            else:
                segs = ([[PHASE_D_LOSS, PHASE_D_GP, PHASE_D_STEP, PHASE_G_LOSS, PHASE_G_STEP]], [], [])
            kind = (("u8", real.data.data_ptr()) if isinstance(real, DeviceBatch) else ("f32",)) + (nz.get("n1") is None, nz.get("z") is None, nz.get("alpha") is None, nz.get("m1") is None)
            handle = None

            launched = []

            def run(seg_id, phases):
                key = (seg_id, step & 1, kind, float(grad_scale), tuple(phases))
                ge = self._graph_cache.get(key)
                if ge is None:
                    global _CAPTURES_OPEN
                    import gc
                    gc_was = gc.isenabled()
                    gc.disable()            # no destructor of unrelated objects (tensors, events, engines) inside the capture
                    try:
                        lib.jck_engine_capture_begin(h, st)
                        _CAPTURES_OPEN += 1
                        try:
                            for ph in phases:
                                lib.jck_engine_phase(h, ph, C.byref(si), st)
                            out = C.c_void_p()
                            lib.jck_engine_capture_end(h, st, C.byref(out))
                        except Exception:
                            lib.jck_engine_capture_abort(h, st)
                            raise
                        finally:
                            _CAPTURES_OPEN -= 1
                            _flush_deferred()
                    except JckError as e:
                        if gc_was:
                            gc.enable()
                        if launched:        # part of the step already ran: no clean fallback
                            raise
                        raise _GraphUnavailable(str(e))
                    if gc_was:
                        gc.enable()
                    ge = self._graph_cache[key] = out.value
                lib.jck_graph_launch(ge, st)
                launched.append(seg_id)

            run(0, segs[0][0])
            if reduce_d or reduce_g:
                handle = reduce_d(self.arenas["d_grads"]) if reduce_d else None    # no early bucket under replay: its event lives in the graph
                if handle is not None:
                    handle()
                run(1, segs[1][0])
                handle = reduce_g(self.arenas["g_grads"]) if reduce_g else None
                if handle is not None:
                    handle()
                run(2, segs[2][0])
        self.t += 1
        self._shared["last_step"] = self.t
        self._shared["version"] += 1
        self._packed_version = self._shared["version"]
        self._keep = keep

    def step_async(self, real, noise=None, lr=2e-4, reduce_d=None, reduce_g=None, grad_scale=1.0, graph=None,
                   generator=None, labels=None, next_real=None, next_noise=None, lr_g=None):
        """Enqueues one full step; no host sync.  noise=None draws on the device (generator= / labels= as draw_noise takes
        them).  lr_g: G's learning rate when it differs from D's `lr` (eager launches only).  graph (default: off, env
        JCK_GRAPH=1 enables; never with per-launch profiling): replay the step from captured hipGraphs on the engine's own
        stream - the first step of an engine always runs eagerly.  `reduce_d/reduce_g(flat_grads)` are called between the loss
        and the optimiser phases (data-parallel gradient all-reduce) and return a wait-callable.  With self.ddp_overlap
        (default; JCK_DDP_SPLIT=0 or hipgan.dist.ReplicaGuard switch it off) the DCGAN step puts D's all-reduce in two pieces
        under D's own backward, and:
        next_real (DCGAN, batched schedule, eager launches): the NEXT step's real batch.  The forward half of
        its D(real) pass (input transform, instance noise - next_noise["n1"] when the caller supplies noise tensors -, conv
        stack + BatchNorm statistics) is then enqueued right behind the start of G's gradient all-reduce, so the collective
        runs under ~0.13 ms of compute that needs no G weights, instead of being waited for at once; the next step_async call
        must be given that same batch.  Results are bitwise those of the plain order (PHASE_D_REAL_FWD, include/jckgan.h)."""
        if self._packed_version != self._shared["version"]:
            self.join()
            self.repack()
        use_graph = (self.graphs if graph is None else graph) and self._eager_steps >= 1 and lr_g is None
        pre_key = getattr(self, "_prefetched_real", None)
        if pre_key is not None:
            self._prefetched_real = None
            if pre_key != self._real_key(real):
                raise JckError("step_async: the batch announced as next_real of the previous step must be this step's real batch")
            use_graph = False                       # D(real)'s forward of this step is already enqueued
        if use_graph:
            try:
                self._fallback_inputs = None
                return self._step_graph(real, noise, lr, reduce_d, reduce_g, grad_scale, generator, labels)
            except _GraphUnavailable as e:
                import warnings
                warnings.warn(f"hipGraph replay disabled for this engine ({e}); running the step eagerly")
                self.graphs = False
                torch.cuda.current_stream().wait_stream(self._e_stream())
                if self._fallback_inputs is not None:      # this step's inputs are already drawn / copied into the static buffers
                    real, noise = self._fallback_inputs
                    self._fallback_inputs = None
        self._eager_steps += 1
        noise = noise if noise is not None else self.draw_noise(generator, labels=labels, fast=self.fast_noise)
        si, keep = self._inputs(real, noise, lr, grad_scale)
        st = torch.cuda.current_stream().cuda_stream
        h = self._h
        self.join()
        # phases issued between the start of an all-reduce and the wait for it take the three-launch BatchNorm backward when the
        # collective has peers to wait for (N > 1): a resident launch needs every CU and RCCL's kernel holds some
        nores = PHASE_NO_RESIDENT if ((reduce_d or reduce_g) and self.collective_world() > 1) else 0
        tail = int(lib.jck_engine_grad_tail(h, 1)) if (reduce_d and self.family == 0 and self.ddp_overlap) else -1
        if tail > 0:
            # data parallel, batched schedule: the tail of D's gradient arena (conv4.weight .. conv5.weight, 76 % of its
            # bytes) is final after the first weight-gradient product of the backward pass - its all-reduce is started
            # there, in plain stream order, and runs under the remaining ~0.5 ms of the pass; the head follows at the end
            flat = self.arenas["d_grads"]
            if self.ddp_lazy_tail:
                # the phase does not make this stream wait for the weight-gradient stream (that stall cost more than the early
                # collective hid): the tail's all-reduce is issued from a stream of its own that waits for the tail's last writers
                lib.jck_engine_phase(h, PHASE_D_LOSS_A | PHASE_LAZY_JOIN, C.byref(si), st)
                if self._ar_stream is None:
                    self._ar_stream = torch.cuda.Stream()
                lib.jck_engine_order_after_tail(h, self._ar_stream.cuda_stream)
                with torch.cuda.stream(self._ar_stream):
                    w_tail = reduce_d(flat[tail:])
            else:
                lib.jck_engine_phase(h, PHASE_D_LOSS_A, C.byref(si), st)
                w_tail = reduce_d(flat[tail:])
            lib.jck_engine_phase(h, PHASE_D_LOSS_B | nores, C.byref(si), st)      # the tail's all-reduce is in flight
            w_head = reduce_d(flat[:tail])
            lib.jck_engine_phase(h, PHASE_D_GP | nores, C.byref(si), st)
            for w in (w_tail, w_head):
                if w is not None:
                    w()
        else:
            # CGAN without an all-reduce in between: nothing reads D's gradients before PHASE_D_STEP, which then closes the join
            # with the weight-gradient stream itself (include/jckgan.h: JCK_PHASE_LAZY_JOIN)
            lazy = PHASE_LAZY_JOIN if (self.family == 1 and not reduce_d and self.lazy_join) else 0
            lib.jck_engine_phase(h, PHASE_D_LOSS | lazy, C.byref(si), st)
            if self.family == 0:
                handle = reduce_d(self.arenas["d_grads"]) if reduce_d else None
                lib.jck_engine_phase(h, PHASE_D_GP | (nores if handle is not None else 0), C.byref(si), st)   # the penalty pass overlaps the D all-reduce (no gradients)
            else:                                                      # CGAN back-propagates the penalty: reduce after it
                lib.jck_engine_phase(h, PHASE_D_GP | lazy, C.byref(si), st)
                handle = reduce_d(self.arenas["d_grads"]) if reduce_d else None
            if handle is not None:
                handle()
        lib.jck_engine_phase(h, PHASE_D_STEP, C.byref(si), st)
        lib.jck_engine_phase(h, PHASE_G_LOSS, C.byref(si), st)
        if lr_g is not None:
            si.lr = lr_g                 # the engine rewrites only the Adam scalars of the step for the new rate
        handle = reduce_g(self.arenas["g_grads"]) if reduce_g else None
        # the forward half of the next step's D(real) pass, announced by the caller: under G's all-reduce when data parallel, and on
        # one GPU beside this step's Adam(G) + repack and the next step's set-step launch, which leave the second stream idle
        # (round 5: -0.4 % of the step; JCK_PREFETCH_SINGLE=0 keeps it for the data-parallel step only)
        if (next_real is not None and self.family == 0 and self.ddp_overlap and getattr(self, "_prefetch_ok", True)
                and not (self.graphs if graph is None else graph)
                and (handle is not None or os.environ.get("JCK_PREFETCH_SINGLE", "1") != "0")):
            keep += self._prefetch_real(next_real, next_noise, lr, grad_scale, st)
        if handle is not None:
            handle()
        lib.jck_engine_phase(h, PHASE_G_STEP, C.byref(si), st)
        self.t += 1
        self._t_engine, self._module_epoch = self.t, getattr(self, "_module_epoch", 0) + 1      # (hipgan/optim.py: EngineAdam.step)
        self._shared["last_step"] = self.t
        self._shared["version"] += 1            # weights moved; this engine's packs were refreshed by the step itself
        self._packed_version = self._shared["version"]
        self._keep = keep

    @staticmethod
    def _real_key(real):
        return (real.data.data_ptr(), real.idx.data_ptr()) if isinstance(real, DeviceBatch) else (real.data_ptr(),)

    def _prefetch_real(self, next_real, next_noise, lr, grad_scale, st):
        """PHASE_D_REAL_FWD of step t+2 (self.t is still t during step t+1).  -> tensors to keep alive until the stream has run."""
        si2, keep2 = self._inputs(next_real, {"n1": (next_noise or {}).get("n1")}, lr, grad_scale)
        si2.step = self.t + 2
        try:
            lib.jck_engine_phase(self._h, PHASE_D_REAL_FWD, C.byref(si2), st)
        except JckError:
            self._prefetch_ok = False               # per-pass schedule (batch % 8 != 0, JCK_BATCHED): no such split
            return []
        self._prefetched_real = self._real_key(next_real)
        return keep2

    def record_scalars(self, dst_row):
        """Copies the step scalars (device float[8]) into `dst_row` on the stream that produced them - no host sync and no
        stall of the next step's D pass."""
        gs = self._shared.get("e_stream")
        src = self.scalars_view(joined=False)
        if gs is None:
            dst_row.copy_(src, non_blocking=True)
        else:
            gs.wait_stream(torch.cuda.current_stream())          # dst_row may have been produced on the caller's stream
            with torch.cuda.stream(gs):
                dst_row.copy_(src, non_blocking=True)

    def scalars_view(self, joined=True):
        """Device view (float32[8]) of the last step's scalars - no host sync (the current stream joins the G phase)."""
        if joined:
            self.join()
        return self._ws_view(load_library().jck_engine_scalars_at(self._h, self._shared["last_step"]), 8, torch.float32)

    def scalars(self):
        """Host copy of the eight step scalars (one device->host sync)."""
        vals = self.scalars_view().cpu().tolist()
        self.check()
        return dict(zip(SCALAR_NAMES, vals))

    def gradient_penalty_pass(self, real, fake, alpha, labels=None, drop_mask=None):
        """PHASE_GP_ONLY on this engine's D weights: -> per-image gradient norms [B] (device fp32 view, valid until the next
        call).  CGAN engines also leave d(penalty)/d(theta_D) in arenas["d_grads"] (cleared first).  real / fake: [B,3,S,S]
        fp32 device tensors taken as they are, alpha [B]; CGAN: labels int64 one-hot [B,100], drop_mask float keep-mask [B,256]."""
        B, S = self.batch, self.size
        self.join()
        if self._packed_version != self._shared["version"]:
            self.repack()
        si, keep = StepInputs(), []

        def f32(t, n, what):
            t = t.detach().to(self.device, torch.float32).contiguous()
            if t.numel() != n:
                raise JckError(f"gradient_penalty_pass: {what} has {t.numel()} elements, expected {n}")
            keep.append(t)
            return t.data_ptr()
        si.real_nchw, si.noise_real = f32(real, B * 3 * S * S, "real"), f32(fake, B * 3 * S * S, "fake")
        si.alpha = f32(alpha, B, "alpha")
        si.lr, si.grad_scale, si.step = 2e-4, 1.0, self.t + 1
        if self.family == 1:
            if labels is None or tuple(labels.shape) != (B, 100) or drop_mask is None:
                raise JckError("CGAN gradient penalty needs labels (int64 one-hot [B,100]) and a dropout keep-mask [B,256]")
            lab = labels.to(self.device, torch.int64).contiguous()
            keep.append(lab)
            si.labels = lab.data_ptr()
            si.drop_mask[2] = f32(drop_mask, B * 256, "drop_mask")
        lib.jck_engine_phase(self._h, PHASE_GP_ONLY, C.byref(si), torch.cuda.current_stream().cuda_stream)
        self._keep = keep
        n = C.c_longlong()
        p = load_library().jck_engine_tensor(self._h, b"norms", C.byref(n))
        return self._ws_view(p, n.value, torch.float32)[:B]

    def check(self):
        """Raises JckError if a grid barrier of a resident launch timed out since the last call (the step's results are invalid;
        the optimiser phases left the weights untouched meanwhile).  Call behind a host synchronisation: the step scalars, the
        replica guard, checkpoint / evaluation snapshots, the end of training and bench.py do."""
        lib.jck_engine_check(self._h)

    @staticmethod
    def collective_world():
        """Ranks a gradient all-reduce waits for: torch.distributed's world size, 1 without a process group."""
        import torch.distributed as dist
        w = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        # JCK_ASSUME_WORLD (measurement hook): schedule the step as a rank of that many would - no grid-barrier launch under a
        # collective - with the ranks actually present, to price that schedule on one device
        return max(w, int(os.environ.get("JCK_ASSUME_WORLD", "1")))

    def _ws_view(self, ptr, numel, dtype):
        """Typed view of a region of the bound workspace given its device address."""
        off = ptr - self.workspace.data_ptr()
        nbytes = numel * torch.empty(0, dtype=dtype).element_size()
        if off < 0 or off + nbytes > self.ws_bytes:
            raise JckError("pointer outside the engine workspace")
        return self.workspace[off:off + nbytes].view(dtype)

    def step(self, real, noise=None, lr=2e-4, **kw):
        self.step_async(real, noise, lr, **kw)
        return self.scalars()

    def sample(self, z, labels=None):
        """G(z) with train-mode BatchNorm (train/dcgan_trainer.py:199-200) -> NCHW fp32 on the device.  The whole z is ONE
        BatchNorm batch (statistics and running-stat update over all n samples, as in the reference), so n <= batch."""
        n = z.shape[0]
        if n > self.batch:
            raise JckError(f"sample: {n} latent vectors exceed this engine's batch {self.batch}; bind an engine with batch >= n "
                           f"(DcganEngine(batch=n, share=engine))")

        self.join()
        if self._packed_version != self._shared["version"]:
            self.repack()
        out = torch.empty(n, 3, self.size, self.size, dtype=torch.float32, device=self.device)
        zc = z.to(self.device, torch.float32).contiguous().view(-1, 100)
        lab = None
        if self.family == 1:
            if labels is None or labels.shape != (n, 100):
                raise JckError("CGAN sample needs one-hot int64 labels [n,100]")
            lab = labels.to(self.device, torch.int64).contiguous()
        lib.jck_engine_sample(self._h, zc, lab, n, out, cur_stream())
        self._keep_z = (zc, lab)
        return out

    def tensor(self, name):
        """Debug/parity view of an internal NHWC tensor as a torch tensor (copy)."""
        self.join()
        n = C.c_longlong()
        p = load_library().jck_engine_tensor(self._h, name.encode(), C.byref(n))
        if not p:
            raise KeyError(name)
        f32 = name in ("prob", "ds", "norms", "acc", "rs", "prob_gp")
        dt = torch.float32 if (f32 or self.prec == PREC_F32) else torch.bfloat16
        return self._ws_view(p, n.value, dt).clone()


class CganEngine(DcganEngine):
    """Conditional GAN (model/CGAN.py, train/cgan_trainer.py:173-213): labels + dropout masks ride in the noise dict
    (`labels` int64 one-hot [B,100], `m1..m4` float keep-masks [B,256]); the gradient penalty is back-propagated."""

    family = 1
