"""Host-side mirror of the reference's network module (utils/DSen2Net.py) over libdsen2_hip.so.

``s2model(input_shape, num_layers, feature_size)`` keeps the reference's name, arguments and defaults
(utils/DSen2Net.py:18) and returns an object with the two keras.Model methods the inference path uses:
``load_weights(path)`` (testing/supres.py:63) and ``predict(list_of_arrays, verbose=...)`` (supres.py:65).
All arithmetic happens in the HIP kernels behind the C ABI; PyTorch-ROCm only provides device buffers,
H2D/D2H copies and the stream.
"""
import ctypes
import sys
import threading

import numpy as np
import torch

from . import _lib, weights as _weights

RES_SCALE = 0.1   # resBlock(scale=0.1), utils/DSen2Net.py:9
# arithmetic of the residual-block convolutions (include/dsen2_hip.h: dsen2_model_create).  'fp32' is what keras computes and
# the default everywhere; 'bf16' = bf16 operands (~1e-3 relative error); 'bf16x3' = every fp32 operand as two bf16 numbers,
# three bf16 MFMAs per product (~1e-5 whole-network rmse, inside the 1e-4 gate; ~3 x the fp32 rate)
PRECISIONS = {'fp32': 0, 'bf16': 1, 'bf16x3': 2}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class S2Model(object):
    """What keras' Model is to the reference: built by s2model(), then load_weights() and predict()."""

    def __init__(self, input_shape, num_layers, feature_size, device=None, precision='fp32'):
        if len(input_shape) not in (2, 3):
            raise ValueError('input_shape must describe 2 or 3 inputs, got %r' % (input_shape,))
        if not torch.cuda.is_available():
            raise RuntimeError('dsen2_amd needs a ROCm GPU (gfx950); there is no CPU fallback')
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.bands = tuple(int(s[0]) for s in input_shape)
        self.num_layers = int(num_layers)
        self.feature_size = int(feature_size)
        self.cin = sum(self.bands)
        self.cout = self.bands[-1]          # utils/DSen2Net.py:35 — input_shape[-1][0]
        self._handle = ctypes.c_void_p(0)
        c60 = self.bands[2] if len(self.bands) == 3 else 0
        with torch.cuda.device(self.device):
            if precision not in PRECISIONS:
                raise ValueError("precision must be one of %s" % ', '.join(repr(k) for k in PRECISIONS))
            self.precision = precision
            _lib.call('dsen2_model_create', ctypes.byref(self._handle), self.bands[0], self.bands[1], c60,
                      self.num_layers, self.feature_size, PRECISIONS[precision])
        # One workspace per STREAM the model is used from (keyed by the stream's handle): SURVEY §8(b) — "calls on a
        # handle are serialised by the given stream" — so forwards enqueued on different streams, from one thread or
        # several, must not share the activation buffers the kernels of both would be writing.
        self._workspaces = {}
        self._ws_lock = threading.Lock()
        self.max_workspace_bytes = 6 << 30   # predict() and the tile path size their batches to stay below this

    # -- keras.Model surface -------------------------------------------------------------------
    def count_params(self):
        return int(_lib.load().dsen2_model_num_params(self._handle))

    def set_weights_flat(self, flat):
        flat = np.ascontiguousarray(flat, np.float32).ravel()
        with torch.cuda.device(self.device):
            _lib.call('dsen2_model_load_weights', self._handle,
                      flat.ctypes.data_as(_lib.c_float_p), flat.size)

    def load_weights(self, path):
        self.set_weights_flat(_weights.load_flat(path, self.cin, self.cout, self.num_layers, self.feature_size))

    def workspace_bytes(self, n, h, w):
        out = ctypes.c_size_t(0)
        _lib.call('dsen2_model_workspace_bytes', self._handle, n, h, w, ctypes.byref(out))
        return out.value

    def _get_workspace(self, nbytes):
        """The current stream's workspace, grown on demand.  Allocated under that stream, so torch's caching allocator
        hands a replaced (smaller) buffer back only to work ordered after the kernels still reading it."""
        key = torch.cuda.current_stream(self.device).cuda_stream
        with self._ws_lock:
            ws = self._workspaces.get(key)
            if ws is None or ws.numel() < nbytes:
                self._workspaces.pop(key, None)
                ws = None
                ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
                self._workspaces[key] = ws
            return ws

    def release_workspaces(self):
        """Drop every stream's activation buffers (they are re-created on the next forward)."""
        with self._ws_lock:
            self._workspaces.clear()

    def forward_device(self, xs, out=None):
        """One batch entirely on the device: xs = list of contiguous float32 CUDA tensors [n,c,h,w]."""
        if len(xs) != len(self.bands):
            raise ValueError('expected %d inputs, got %d' % (len(self.bands), len(xs)))
        n, _, h, w = xs[0].shape
        for x, c in zip(xs, self.bands):
            if tuple(x.shape) != (n, c, h, w) or x.dtype != torch.float32 or not x.is_contiguous() or \
                    x.device != self.device:
                raise ValueError('input must be a contiguous float32 %s tensor of shape %r, got %r %s %s'
                                 % (self.device, (n, c, h, w), tuple(x.shape), x.dtype, x.device))
        out = self._check_out(out, n, h, w)
        if n == 0:
            return out
        ws = self._get_workspace(self.workspace_bytes(n, h, w))
        with torch.cuda.device(self.device):
            _lib.call('dsen2_model_forward', self._handle, _ptr(xs[0]), _ptr(xs[1]),
                      _ptr(xs[2]) if len(xs) == 3 else ctypes.c_void_p(0), _ptr(out), n, h, w,
                      _ptr(ws), ws.numel(), _stream_ptr(self.device))
        return out

    def _check_out(self, out, n, h, w):
        """The raw pointer of `out` goes straight to the kernels: a wrong-sized, strided or other-device tensor
        would be written out of bounds or in the wrong place."""
        if out is None:
            return torch.empty((n, self.cout, h, w), dtype=torch.float32, device=self.device)
        if tuple(out.shape) != (n, self.cout, h, w) or out.dtype != torch.float32 or not out.is_contiguous() or \
                out.device != self.device:
            raise ValueError('out must be a contiguous float32 %s tensor of shape %r, got %r %s %s (contiguous: %s)'
                             % (self.device, (n, self.cout, h, w), tuple(out.shape), out.dtype, out.device,
                                out.is_contiguous()))
        return out

    def time_body_in_forward(self, xs, out=None, iters=10):
        """Mean duration (ms) of ONE residual-block convolution launch inside `iters` full forward passes on `xs`
        (HIP events on the launch stream around the 2*num_layers body convolutions of each pass): the duration the
        kernel has in the running network — bench.py's roofline measurement."""
        n, _, h, w = xs[0].shape
        out = self._check_out(out, n, h, w)
        ws = self._get_workspace(self.workspace_bytes(n, h, w))
        ms = ctypes.c_float(0)
        with torch.cuda.device(self.device):
            _lib.call('dsen2_model_forward_timed', self._handle, _ptr(xs[0]), _ptr(xs[1]),
                      _ptr(xs[2]) if len(xs) == 3 else ctypes.c_void_p(0), _ptr(out), n, h, w,
                      _ptr(ws), ws.numel(), _stream_ptr(self.device), int(iters), ctypes.byref(ms))
        return ms.value

    def profile_forward(self, xs, out=None, iters=10, warm=3):
        """`warm` plain forward passes, then — the stream never idling in between — `iters` passes on `xs` with four HIP
        events each (dsen2_model_forward_profile).  Mean milliseconds: forward_ms (first event to last of a pass), first_ms /
        body_ms / out_ms (first convolution, all 2*num_layers residual-block convolutions, output convolution: consecutive
        intervals, they add up to forward_ms) and wall_ms (per instrumented pass, first event of the first pass to last event
        of the last: a pass with its event records and the gap to the next one)."""
        n, _, h, w = xs[0].shape
        out = self._check_out(out, n, h, w)
        ws = self._get_workspace(self.workspace_bytes(n, h, w))
        ms = (ctypes.c_float * 5)()
        with torch.cuda.device(self.device):
            _lib.call('dsen2_model_forward_profile', self._handle, _ptr(xs[0]), _ptr(xs[1]),
                      _ptr(xs[2]) if len(xs) == 3 else ctypes.c_void_p(0), _ptr(out), n, h, w,
                      _ptr(ws), ws.numel(), _stream_ptr(self.device), int(warm), int(iters), ms)
        return dict(forward_ms=ms[0], first_ms=ms[1], body_ms=ms[2], out_ms=ms[3], wall_ms=ms[4])

    def body_launches(self, n, h, w):
        """Kernel launches the 2*num_layers residual-block convolutions of a batch take: 1 = one chain launch."""
        with torch.cuda.device(self.device):
            r = _lib.load().dsen2_model_body_launches(self._handle, int(n), int(h), int(w))
        if r < 0:
            raise _lib.DSen2Error(r, _lib.load().dsen2_last_error().decode())
        return r

    def batch_limit(self, h, w):
        per = self.workspace_bytes(1, h, w)
        return max(1, int(self.max_workspace_bytes // per))

    def preferred_batch(self, h, w):
        """The batch the tile path and predict() cut their work into: batch_limit() for fp32 and for large patches; in the
        bf16 modes, for patches the chain kernel takes (up to 64 x 64: there it is 1-3 % ahead of the per-layer launches and
        bit-identical to them; profiles/r04_k_chain_vs_layerwise.txt), the largest batch below the limit that runs ALL
        residual-block convolutions in one chain launch (body_launches() == 1: a multiple of the CU count).  Results never
        depend on the batch size."""
        limit = self.batch_limit(h, w)
        if self.precision == 'fp32' or self.num_layers <= 0:
            return limit
        cus = int(torch.cuda.get_device_properties(self.device).multi_processor_count)
        n = (limit // cus) * cus
        while n >= cus:
            if self.body_launches(n, h, w) == 1:
                return n
            n -= cus
        return limit

    def predict(self, x, batch_size=None, verbose=0):
        """keras Model.predict: list of NCHW float32 ndarrays -> ndarray [N, cout, H, W].

        ``batch_size`` only bounds device memory (results do not depend on it); default: as many
        patches as fit ``max_workspace_bytes``.  Host<->device copies go through page-locked buffers on their own
        streams; with more than one batch, batch i+1 is staged and batch i-1 downloaded while batch i computes.
        Activation buffers are per stream: a model may be used from several streams / threads at once (the weights
        are read-only after load_weights).
        """
        xs = [np.ascontiguousarray(a, dtype=np.float32) for a in x]
        if len(xs) != len(self.bands):
            raise ValueError('expected %d inputs, got %d' % (len(self.bands), len(xs)))
        n, _, h, w = xs[0].shape
        for a, c in zip(xs, self.bands):
            if a.shape != (n, c, h, w):
                raise ValueError('input of shape %r where %r is expected' % (a.shape, (n, c, h, w)))
        bs = self.preferred_batch(h, w) if batch_size is None else int(batch_size)
        out = np.empty((n, self.cout, h, w), np.float32)
        if n == 0:
            return self._progress_end(verbose, out)
        starts = list(range(0, n, bs))
        bs = min(bs, n)
        with torch.cuda.device(self.device):
            comp = torch.cuda.current_stream(self.device)
            h2d, d2h = torch.cuda.Stream(self.device), torch.cuda.Stream(self.device)
            # one slot for a single batch, two to overlap staging / compute / download of consecutive batches; every
            # transfer goes through page-locked memory on its own stream (a pageable copy blocks the host and runs at
            # a fifth of the PCIe rate)
            slots = []
            for _ in range(min(2, len(starts))):
                slots.append(dict(
                    pin_in=[torch.empty((bs, c, h, w), dtype=torch.float32, pin_memory=True) for c in self.bands],
                    dev_in=[torch.empty((bs, c, h, w), dtype=torch.float32, device=self.device) for c in self.bands],
                    dev_out=torch.empty((bs, self.cout, h, w), dtype=torch.float32, device=self.device),
                    pin_out=torch.empty((bs, self.cout, h, w), dtype=torch.float32, pin_memory=True),
                    ev_in=torch.cuda.Event(), ev_comp=torch.cuda.Event(), ev_out=torch.cuda.Event(), span=None))

            def collect(slot):                      # batch whose download was issued from this slot
                if slot['span'] is not None:
                    slot['ev_out'].synchronize()
                    j0, j1 = slot['span']
                    out[j0:j1] = slot['pin_out'][:j1 - j0].numpy()
                    slot['span'] = None
                    self._progress(verbose, j1, n)

            for i, i0 in enumerate(starts):
                i1 = min(n, i0 + bs)
                m = i1 - i0
                slot = slots[i % len(slots)]
                collect(slot)                       # batch i-2: after this the slot's buffers are all free
                for k, a in enumerate(xs):          # host copy into page-locked memory, under batch i-1's compute
                    slot['pin_in'][k][:m].numpy()[...] = a[i0:i1]
                with torch.cuda.stream(h2d):
                    for k in range(len(xs)):
                        slot['dev_in'][k][:m].copy_(slot['pin_in'][k][:m], non_blocking=True)
                    slot['ev_in'].record(h2d)
                comp.wait_event(slot['ev_in'])
                self.forward_device([t[:m] for t in slot['dev_in']], out=slot['dev_out'][:m])
                slot['ev_comp'].record(comp)
                with torch.cuda.stream(d2h):
                    d2h.wait_event(slot['ev_comp'])
                    slot['pin_out'][:m].copy_(slot['dev_out'][:m], non_blocking=True)
                    slot['ev_out'].record(d2h)
                slot['span'] = (i0, i1)
            last = len(starts) - 1
            if len(slots) > 1:
                collect(slots[(last - 1) % 2])
            collect(slots[last % len(slots)])
        return self._progress_end(verbose, out)

    @staticmethod
    def _progress(verbose, done, total):
        if verbose:
            sys.stdout.write('\r%d/%d' % (done, total))
            sys.stdout.flush()

    @staticmethod
    def _progress_end(verbose, out):
        if verbose:
            sys.stdout.write('\n')
        return out

    def time_body_conv(self, layer, x_in, aux, out, iters=10):
        """Mean duration (ms) of `iters` launches of body convolution `layer` (1-based), HIP events on the
        launch stream — bench.py's roofline measurement."""
        n, h, w, _ = x_in.shape
        ms = ctypes.c_float(0)
        with torch.cuda.device(self.device):
            _lib.call('dsen2_model_time_body_conv', self._handle, layer, _ptr(x_in), _ptr(aux), _ptr(out), n, h, w,
                      iters, _stream_ptr(self.device), ctypes.byref(ms))
        return ms.value

    def __del__(self):
        try:
            if self._handle:
                _lib.load().dsen2_model_destroy(self._handle)
                self._handle = ctypes.c_void_p(0)
        except Exception:
            pass


def s2model(input_shape, num_layers=32, feature_size=256, device=None, precision='fp32'):
    """utils/DSen2Net.py:18 — same positional arguments and defaults.  precision='bf16' / 'bf16x3' run the residual-block
    convolutions on the bf16 matrix cores (fp32 accumulate, exact fp32 residual stream; PRECISIONS above)."""
    return S2Model(input_shape, num_layers, feature_size, device=device, precision=precision)


def to_blocked(x_nhwc):
    """[n,h,w,c] -> the blocked layout of the bf16 kernels [n, c/8, h, w, 8] (a torch reshuffle, test helper)."""
    n, h, w, c = x_nhwc.shape
    return x_nhwc.reshape(n, h, w, c // 8, 8).permute(0, 3, 1, 2, 4).contiguous()


def from_blocked(x_blk):
    n, b, h, w, e = x_blk.shape
    return x_blk.permute(0, 2, 3, 1, 4).reshape(n, h, w, b * e).contiguous()


def split_f32(x):
    """fp32 NHWC CUDA tensor -> (hi, lo): blocked int16 tensors [n, c/8, h, w, 8] (include/dsen2_hip.h: dsen2_split_f32)."""
    x = x.contiguous()
    n, h, w, c = x.shape
    hi = torch.empty((n, c // 8, h, w, 8), dtype=torch.int16, device=x.device)
    lo = torch.empty_like(hi)
    with torch.cuda.device(x.device):
        _lib.call('dsen2_split_f32', _ptr(x), _ptr(hi), _ptr(lo), n, h, w, c, _stream_ptr(x.device))
    return hi, lo


def join_f32(hi, lo):
    n, b, h, w, e = hi.shape
    out = torch.empty((n, h, w, b * e), dtype=torch.float32, device=hi.device)
    with torch.cuda.device(hi.device):
        _lib.call('dsen2_join_f32', _ptr(hi), _ptr(lo), _ptr(out), n, h, w, b * e, _stream_ptr(hi.device))
    return out


def split3_f32(x):
    """fp32 NHWC CUDA tensor -> (hx, lo16): hx int16 [n, 2, c/8, h, w, 8] (plane 0 = hi, plane 1 = xl = bf16(x - hi)) and the
    low halves int16 [n, c/8, h, w, 8] (include/dsen2_hip.h: dsen2_split3_f32) — the residual stream of a 'bf16x3' model."""
    x = x.contiguous()
    n, h, w, c = x.shape
    hx = torch.empty((n, 2, c // 8, h, w, 8), dtype=torch.int16, device=x.device)
    lo = torch.empty((n, c // 8, h, w, 8), dtype=torch.int16, device=x.device)
    with torch.cuda.device(x.device):
        _lib.call('dsen2_split3_f32', _ptr(x), _ptr(hx), _ptr(lo), n, h, w, c, _stream_ptr(x.device))
    return hx, lo


def conv3x3_first_planes(xs, kernel_hwio, bias, precision):
    """Kernel-level entry point of the first convolution of a 'bf16' (precision 1) / 'bf16x3' (2) model on the bf16 matrix
    cores (include/dsen2_hip.h: dsen2_conv3x3_first_planes).  xs: the two or three NCHW float32 CUDA inputs (4 + 6 (+ 2)
    bands); kernel_hwio (3, 3, 10 | 12, feat).  Returns the residual stream as the model holds it: precision 1 (hi, lo) int16
    [n, feat/8, h, w, 8]; precision 2 (hx, lo16) with hx int16 [n, 2, feat/8, h, w, 8] (plane 0 = hi, plane 1 = xl)."""
    kernel_hwio = np.ascontiguousarray(kernel_hwio, np.float32)
    bias = np.ascontiguousarray(bias, np.float32)
    feat = kernel_hwio.shape[3]
    n, _, h, w = xs[0].shape
    dev = xs[0].device
    xs = [x.contiguous() for x in xs]
    if precision == 2:
        out = torch.empty((n, 2, feat // 8, h, w, 8), dtype=torch.int16, device=dev)
    else:
        out = torch.empty((n, feat // 8, h, w, 8), dtype=torch.int16, device=dev)
    out2 = torch.empty((n, feat // 8, h, w, 8), dtype=torch.int16, device=dev)
    with torch.cuda.device(dev):
        _lib.call('dsen2_conv3x3_first_planes', _ptr(xs[0]), _ptr(xs[1]), _ptr(xs[2]) if len(xs) == 3 else ctypes.c_void_p(0),
                  xs[0].shape[1], xs[1].shape[1], xs[2].shape[1] if len(xs) == 3 else 0,
                  kernel_hwio.ctypes.data_as(_lib.c_float_p), bias.ctypes.data_as(_lib.c_float_p), int(feat), int(precision),
                  _ptr(out), _ptr(out2), n, h, w, _stream_ptr(dev))
    return out, out2


def conv3x3_body_bf16x3(x_planes, kernel_hwio, bias, epilogue=0, res_hx=None, res_lo=None, res_scale=RES_SCALE):
    """Kernel-level entry point of the bf16x3 body convolution.  x_planes: int16 (bf16 bit patterns) [n, 2, feat/8, h, w, 8].
    epilogue 0: returns relu(conv + bias) as such a two-plane tensor.  epilogue 1: updates the stream (res_hx, res_lo; see
    split3_f32) in place and returns it.  epilogue 3: returns the updated stream as fp32 NHWC."""
    n, _, blocks, h, w, _ = x_planes.shape
    feat = blocks * 8
    kernel_hwio = np.ascontiguousarray(kernel_hwio, np.float32)
    bias = np.ascontiguousarray(bias, np.float32)
    out = None
    if epilogue == 0:
        out = torch.empty((n, 2, feat // 8, h, w, 8), dtype=torch.int16, device=x_planes.device)
    elif epilogue == 3:
        out = torch.empty((n, h, w, feat), dtype=torch.float32, device=x_planes.device)
    with torch.cuda.device(x_planes.device):
        _lib.call('dsen2_conv3x3_body_bf16x3', _ptr(x_planes.contiguous()), kernel_hwio.ctypes.data_as(_lib.c_float_p),
                  bias.ctypes.data_as(_lib.c_float_p), _ptr(res_hx), _ptr(res_lo), _ptr(out), n, h, w, feat, int(epilogue),
                  float(res_scale), _stream_ptr(x_planes.device))
    return (res_hx, res_lo) if epilogue == 1 else out


def conv3x3_body_bf16(x_bf16, kernel_hwio, bias, epilogue=0, res_hi=None, res_lo=None, res_scale=RES_SCALE):
    """Kernel-level entry point of the bf16 body convolution: x_bf16 NHWC torch.bfloat16 CUDA tensor (reshuffled to
    the kernel's blocked layout here).  epilogue 0: returns relu(conv + bias) as bf16 NHWC.  epilogue 1: updates the
    residual stream's blocked planes (res_hi, res_lo; see split_f32) in place and returns them.  epilogue 3:
    returns the updated residual stream as fp32 NHWC (planes untouched)."""
    n, h, w, feat = x_bf16.shape
    kernel_hwio = np.ascontiguousarray(kernel_hwio, np.float32)
    bias = np.ascontiguousarray(bias, np.float32)
    xb = to_blocked(x_bf16)
    out = None
    if epilogue == 0:
        out = torch.empty((n, feat // 8, h, w, 8), dtype=torch.bfloat16, device=x_bf16.device)
    elif epilogue == 3:
        out = torch.empty((n, h, w, feat), dtype=torch.float32, device=x_bf16.device)
    with torch.cuda.device(x_bf16.device):
        _lib.call('dsen2_conv3x3_body_bf16', _ptr(xb), kernel_hwio.ctypes.data_as(_lib.c_float_p),
                  bias.ctypes.data_as(_lib.c_float_p), _ptr(res_hi), _ptr(res_lo), _ptr(out), n, h, w, feat, int(epilogue),
                  float(res_scale), _stream_ptr(x_bf16.device))
    if epilogue == 1:
        return res_hi, res_lo
    return from_blocked(out) if epilogue == 0 else out


def conv3x3_nhwc(x, kernel_hwio, bias, epilogue=0, aux=None, res_scale=RES_SCALE, ref=False):
    """Single-layer entry point (kernel-level parity tests): x NHWC float32 CUDA tensor.  ref=True runs the
    one-tile-per-workgroup kernel (dsen2_conv3x3_nhwc_ref), the independent structure for cross-checks."""
    n, h, w, cin = x.shape
    kernel_hwio = np.ascontiguousarray(kernel_hwio, np.float32)
    bias = np.ascontiguousarray(bias, np.float32)
    cout = kernel_hwio.shape[3]
    if epilogue == 2:
        out = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
    else:
        out = torch.empty((n, h, w, cout), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.call('dsen2_conv3x3_nhwc_ref' if ref else 'dsen2_conv3x3_nhwc', _ptr(x), kernel_hwio.ctypes.data_as(_lib.c_float_p),
                  bias.ctypes.data_as(_lib.c_float_p), _ptr(aux), _ptr(out), n, h, w, cin, cout, int(epilogue),
                  float(res_scale), _stream_ptr(x.device))
    return out
