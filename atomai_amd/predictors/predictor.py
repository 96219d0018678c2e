"""BasePredictor / SegPredictor (reference: atomai/predictors/predictor.py:23-298).

Same constructor arguments and numpy-in / numpy-out behaviour.  The frame loop of the reference (one
frame per launch, synchronous H2D/D2H, host ``torch.zeros`` of the whole output) is replaced by a
chunked pipeline: pinned staging buffers, H2D / kernels / D2H of consecutive chunks overlapped on HIP
streams.  In eval mode BatchNorm uses running statistics, so frames are independent and the result does
not depend on how the stack is chunked (``num_batches`` keeps its meaning for the caller only).
"""
import time
from typing import List, Tuple, Type, Union

import os

import numpy as np
import torch

from ..nets.fcnn import _HipNet, predict_proba
from .locator import Locator, locate_device
from .. import _lib as L
from ..utils import get_downsample_factor, get_nb_classes, img_pad, img_resize, set_train_rng, torch_format_image


class BasePredictor:
    def __init__(self, model: Type[torch.nn.Module] = None, use_gpu: bool = False, **kwargs) -> None:
        self.model = model
        self.device = "cpu"
        if use_gpu and torch.cuda.is_available():
            self.device = kwargs.get("device") or "cuda"
        if self.model is not None:
            self.model.to(self.device)
        self.verbose = kwargs.get("verbose", False)

    def preprocess(self, data):
        if isinstance(data, np.ndarray):
            data = torch.from_numpy(data).float()
        return data

    def _model2device(self, device: str = None) -> None:
        self.model.to(device or self.device)

    def _data2device(self, data: torch.Tensor, device: str = None) -> torch.Tensor:
        return data.to(device or self.device)

    def forward_(self, xnew: torch.Tensor) -> torch.Tensor:
        self.model.eval()
        with torch.no_grad():
            return self.model(xnew.to(self.device))

    def batch_predict(self, data: torch.Tensor, out_shape: Tuple[int], num_batches: int,
                      on_chunk=None) -> torch.Tensor:
        """Batch-by-batch prediction into a host tensor (predictor.py:82-106).  ``on_chunk(start, out)`` sees
        each chunk's output while it is still on the model's device."""
        bs = max(1, len(data) // max(1, num_batches))
        out = torch.empty(out_shape)
        for i in range(0, len(data), bs):
            res = self.forward_(data[i:i + bs])
            if on_chunk is not None:
                on_chunk(i, res)
            out[i:i + bs] = res.cpu()
        return out

    def predict(self, data: torch.Tensor, out_shape: Tuple[int] = None, num_batches: int = 1):
        out_shape = data.shape if out_shape is None else (data.shape[0], *out_shape)
        return self.batch_predict(self.preprocess(data), out_shape, num_batches)


class _host_threads:
    """Caps torch's intra-op (OpenMP) thread count while the chunk pipeline runs.  On the 128-core MI355X host the
    default pool puts 128 threads into every 64 MB staging copy; after each parallel region they spin, and the ROCm
    runtime's signal-handler thread is scheduled tens of milliseconds late: the host then learns about finished
    downloads only in bursts and the GPU idles one chunk time in three (369 -> 513 frames/s on 256 frames of 1024^2 with
    the cap, tools/gpu_predict_timeline.py).  8 threads move 64 MB in < 1 ms."""

    def __init__(self, n: int = 8):
        self.n = int(os.environ.get("AMX_PREDICT_HOST_THREADS", n))

    def __enter__(self):
        self.old = torch.get_num_threads()
        if 0 < self.n < self.old:
            torch.set_num_threads(self.n)
        return self

    def __exit__(self, *exc):
        if torch.get_num_threads() != self.old:
            torch.set_num_threads(self.old)
        return False


def _min_max(a: np.ndarray):
    """(min, max) of a stack in ONE multi-threaded pass (torch.aminmax): ``a.min()`` + ``np.ptp(a)`` are three
    single-threaded passes — 0.25 s per GB, as long as the MI355X needs to decode the 256 frames of that GB."""
    if a.dtype in (np.float32, np.float64) and a.flags.c_contiguous and a.size:
        mn, mx = torch.aminmax(torch.from_numpy(a))
        return a.dtype.type(mn.item()), a.dtype.type(mx.item())
    return a.min(), a.max()


def _min_ptp(a: np.ndarray):
    """(min, ptp) in the stack's dtype, the two constants of torch_format_image (utils/preproc.py:822-823): min and max
    are exact, and ptp = max - min is the same single rounding numpy's ``np.ptp`` performs."""
    mn, mx = _min_max(a)
    return mn, mx - mn


class SegPredictor(BasePredictor):
    """Prediction with a trained segmentation net (predictor.py:124-298)."""

    def __init__(self, trained_model: Type[torch.nn.Module], refine: bool = False,
                 resize: Union[Tuple, List] = None, use_gpu: bool = False, logits: bool = True,
                 **kwargs: Union[int, float, bool]) -> None:
        super().__init__(trained_model, use_gpu)
        set_train_rng(1)
        self.nb_classes = kwargs.get('nb_classes', None)
        if self.nb_classes is None:
            self.nb_classes = get_nb_classes(trained_model)
        self.downsampling = kwargs.get('downsampling', None)
        if self.downsampling is None:
            self.downsampling = get_downsample_factor(trained_model)
        self.resize = resize
        self.logits = logits
        self.refine = refine
        self.d = kwargs.get("d", None)
        self.thresh = kwargs.get("thresh", .5)
        self.use_gpu = use_gpu
        self.verbose = kwargs.get("verbose", True)
        self.chunk_bytes = int(kwargs.get("chunk_bytes", int(os.environ.get("AMX_PREDICT_CHUNK_MB", "64")) << 20))
        self._norm = None

    def preprocess(self, image_data: np.ndarray, norm: bool = True, device_norm: bool = True) -> torch.Tensor:
        """Pads and formats the stack (predictor.py:190-207).  For a float32 stack the global min-max
        normalisation `(x - min) / ptp` of torch_format_image is NOT applied on the host (4 extra passes over a
        17 GB stack in the reference): only min and ptp are computed here and each chunk is normalised on the
        device right after its upload (`amx_sub_div`, the same two fp32 operations -> identical values)."""
        if image_data.ndim == 2:
            image_data = image_data[np.newaxis, ...]
        elif image_data.ndim == 4:
            if image_data.shape[-1] == 1:
                image_data = image_data[..., 0]
            elif image_data.shape[1] == 1:
                image_data = image_data[:, 0, ...]
        if self.resize is not None:
            image_data = img_resize(image_data, self.resize)  # predictor.py:203-204 (cv2.resize restated on the device)
        image_data = img_pad(image_data, self.downsampling)
        self._norm = None
        on_device = str(self.device).startswith("cuda") or L.is_test_backend()
        if (norm and device_norm and on_device and isinstance(image_data, np.ndarray)
                and image_data.dtype == np.float32 and image_data.ndim == 3):
            fixed = getattr(self, "_fixed_norm", None)       # global (min, ptp) agreed across ranks
            self._norm = fixed if fixed is not None else _min_ptp(image_data)
            return torch.from_numpy(np.ascontiguousarray(image_data[:, None]))
        if norm and getattr(self, "_fixed_norm", None) is not None:
            mn, ptp = self._fixed_norm                       # torch_format_image with the GLOBAL min / ptp
            x = image_data[:, None] if image_data.ndim == 3 else image_data
            return torch.from_numpy(np.ascontiguousarray((x - mn) / ptp)).float()
        return torch_format_image(image_data, norm)

    def forward_(self, images: torch.Tensor) -> torch.Tensor:
        """Probabilities (N,H,W,C) for a batch already on / moved to the model's device."""
        images = images.to(self.device)
        norm = getattr(self, "_norm", None)
        self.model.eval()
        if isinstance(self.model, _HipNet) and self.logits:
            return predict_proba(self.model, images, input_norm=norm)   # normalisation inside the first-layer kernel
        if norm is not None:
            raw = images.contiguous()
            images = torch.empty_like(raw)
            L.call("amx_sub_div", L.ptr(raw), L.ptr(images), raw.numel(), float(norm[0]),
                   float(norm[1]), L.stream_ptr(raw))
        with torch.no_grad():
            prob = self.model(images)
        if self.logits:
            prob = torch.softmax(prob, dim=1) if self.nb_classes > 1 else torch.sigmoid(prob)
        elif self.nb_classes > 1:
            prob = torch.exp(prob)
        return prob.permute(0, 2, 3, 1)

    def batch_predict(self, data: torch.Tensor, out_shape: Tuple[int], num_batches: int,
                      on_chunk=None) -> torch.Tensor:
        if not (torch.cuda.is_available() and str(self.device).startswith("cuda")):
            return super().batch_predict(data, out_shape, num_batches, on_chunk)
        n = len(data)
        out = torch.empty(out_shape)
        per_frame = max(data[0].numel(), int(np.prod(out_shape[1:]))) * 4
        chunk = max(1, min(n, self.chunk_bytes // per_frame))
        dev = torch.device(self.device)
        from ..engine import aux_stream
        copy_in, copy_out = aux_stream(dev, 1), aux_stream(dev, 2)
        main = torch.cuda.current_stream(dev)
        NS = int(os.environ.get("AMX_PREDICT_NS", "3"))   # chunks in flight (host runs up to NS chunks ahead of the GPU)
        # pinned staging buffers are kept on the predictor: allocating 6 x 64 MB of page-locked memory costs more
        # than decoding a hundred frames
        key = (tuple(data.shape[1:]), tuple(out_shape[1:]))
        pinned = getattr(self, "_pinned", None)
        if pinned is None or pinned[0] != key or len(pinned[1][0]) < chunk or len(pinned[1]) != NS:
            self._pinned = (key,
                            [torch.empty((chunk,) + tuple(data.shape[1:]), pin_memory=True) for _ in range(NS)],
                            [torch.empty((chunk,) + tuple(out_shape[1:]), pin_memory=True) for _ in range(NS)])
        pin_in, pin_out = self._pinned[1], self._pinned[2]
        stage = {}                                  # chunk index -> its in-flight state

        # Software pipeline over chunks, NS of them in flight.  The DOWNLOAD is a copy kernel on a side stream that
        # writes straight into pinned host memory: hipMemcpyAsync behind a cross-stream event wait was measured to
        # block the calling thread on this runtime (the "asynchronous" D2H call stalled the host for up to two chunk
        # times and the compute stream ran dry); a kernel launch never blocks.  The UPLOAD stays a hipMemcpyAsync
        # (it depends on nothing, so it is issued without waiting; a copy kernel READING pinned host memory was
        # tried and returned stale data — torch's pinned buffers are not fine-grained coherent).  Tensors stay
        # referenced in `stage` until the streams that read them are known to be done.
        import ctypes

        def finish(k):                              # chunk k queued -> hooks, queue its download
            st = stage[k]
            if on_chunk is not None:
                st["done"].synchronize()
                on_chunk(st["s"], st["prob"])
            prob = st["prob"] = st["prob"].contiguous()
            copy_out.wait_event(st["done"])
            L.call("amx_copy16", L.ptr(prob), ctypes.c_void_p(pin_out[k % NS].data_ptr()), prob.numel() * 4, 128,
                   ctypes.c_void_p(copy_out.cuda_stream))
            st["ev_out"] = torch.cuda.Event()
            st["ev_out"].record(copy_out)

        # Chunk k downloaded -> user-visible output.  This runs on a WORKER thread, so that the launch thread never waits
        # for a download and never spends its time in the 64 MB copy into (first-touched) output pages: at the full size
        # of BASELINE configs[2] (4096 frames: 17 GB in, 17 GB out) the launch thread then needs ~3 ms per 16-frame chunk
        # (staging copy 1.5 ms, upload call 0.3 ms, launches 1 ms) against ~20 ms of kernels and waits for a free slot
        # the rest of the time — the pipeline is GPU-bound (profiles/r03_predict4096.log: 700 frames/s end to end, 720
        # steady, for a device rate of 752; 626-642 -> 682 frames/s on 256 frames against the single-threaded loop).
        import queue
        import threading
        todo: "queue.SimpleQueue" = queue.SimpleQueue()
        slot_free = [threading.Event() for _ in range(NS)]
        for e_ in slot_free:
            e_.set()
        failure = []

        def collector():
            nthr = _host_threads().n
            if 0 < nthr < torch.get_num_threads():
                torch.set_num_threads(nthr)
            while True:
                k = todo.get()
                if k is None:
                    return
                try:
                    st = stage[k]
                    st["ev_out"].synchronize()
                    out[st["s"]:st["s"] + st["m"]] = pin_out[k % NS][:st["m"]]
                    stage.pop(k)                    # its staging buffers and device tensors are free again
                except BaseException as exc:        # surfaced by the main thread
                    failure.append(exc)
                finally:
                    slot_free[k % NS].set()

        worker = threading.Thread(target=collector, name="amx-predict-collect", daemon=True)
        worker.start()

        def wait_slot(slot):
            slot_free[slot].wait()
            if failure:
                raise failure[0]

        nchunks = 0
        trace = [] if os.environ.get("AMX_PREDICT_TRACE") else None      # dev aid: host time per phase and chunk
        import time as _time
        try:
            for k, s in enumerate(range(0, n, chunk)):
                slot, m = k % NS, min(chunk, n - s)
                t0 = _time.perf_counter()
                wait_slot(slot)                     # chunk k - NS is in `out`: pin_in / pin_out[slot] may be reused
                t1 = _time.perf_counter()
                pin_in[slot][:m].copy_(data[s:s + m])   # host memcpy overlaps the GPU work of the chunks in flight
                t2 = _time.perf_counter()
                with torch.cuda.stream(copy_in):    # upload: hipMemcpyAsync with no dependency (see above)
                    d = pin_in[slot][:m].to(dev, non_blocking=True)
                ev_in = torch.cuda.Event()
                ev_in.record(copy_in)
                main.wait_event(ev_in)
                t3 = _time.perf_counter()
                prob = self.forward_(d)
                done = torch.cuda.Event()
                done.record(main)
                stage[k] = {"d": d, "prob": prob, "done": done, "s": s, "m": m}
                finish(k)
                slot_free[slot].clear()
                todo.put(k)
                nchunks = k + 1
                if trace is not None:
                    trace.append((k, t1 - t0, t2 - t1, t3 - t2, _time.perf_counter() - t3))
        finally:
            todo.put(None)
            worker.join()
        if failure:
            raise failure[0]
        if trace:
            for k, a, b, c, d_ in trace:
                print(f"chunk {k:3d}: wait-slot {1e3*a:7.2f}  copy-in {1e3*b:7.2f}  upload {1e3*c:7.2f}  launches {1e3*d_:7.2f} ms")
        return out

    # ------------------------------------------------------------------ multi-GPU
    @staticmethod
    def _dist_world():
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return dist.get_rank(), dist.get_world_size()
        return 0, 1

    @staticmethod
    def frame_range(n: int, rank: int, world: int) -> Tuple[int, int]:
        """Contiguous frame range [lo, hi) of `rank`: frames are independent in eval mode, so the stack shards with
        no data-path collective."""
        return n * rank // world, n * (rank + 1) // world

    def _global_min_ptp(self, local: np.ndarray):
        """Global (min, ptp) of the stack from each rank's LOCAL frames: ONE all-reduce(MAX) of the two floats
        (-min, max) — the only exchange of the predict path (utils/preproc.py:822-823 normalises by the min / ptp of
        the WHOLE stack).  float32 min / max are exact, so the result equals numpy's over the full stack bit for bit."""
        import torch.distributed as dist
        if local.size:
            mn_, mx_ = _min_max(local)
            v = [-float(mn_), float(mx_)]
        else:
            v = [-float("inf"), -float("inf")]
        dev = self.device if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor(v, dtype=torch.float64, device=dev)      # float64 holds float32 / int32 values exactly
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ty = local.dtype.type                                      # min / ptp in the stack's own dtype, as numpy's
        mn, mx = ty(-t[0].item()), ty(t[1].item())
        return mn, mx - mn

    def predict_distributed(self, image_data: np.ndarray, gather: bool = True, **kwargs):
        """``predict`` with the frames of the stack sharded over the ranks of the initialised process group (one
        process per GPU).  Every rank passes the SAME stack (or a memory-map of it) and decodes only its contiguous
        range; normalisation uses the global min / ptp (`_global_min_ptp`).  With ``gather`` rank 0 returns the whole
        decoded stack (other ranks return their own range — possibly empty, shape (0, H, W, C)); without it every rank
        returns (lo, decoded range)."""
        import torch.distributed as dist
        rank, world = self._dist_world()
        if image_data.ndim == 2:
            image_data = image_data[np.newaxis, ...]
        n = len(image_data)
        lo, hi = self.frame_range(n, rank, world)
        local = np.ascontiguousarray(image_data[lo:hi])
        norm = kwargs.get("norm", True)
        fixed = None
        if norm and world > 1:
            fixed = self._global_min_ptp(local)
        self._fixed_norm = fixed
        try:
            mine = self.predict(local, **kwargs) if hi > lo else None
        finally:
            self._fixed_norm = None
        if world == 1:
            return mine if gather else (lo, mine)
        # a rank whose range is empty (more ranks than frames) returns an EMPTY (0, H, W, C) array, never None
        shapes = [None] * world
        dist.all_gather_object(shapes, None if mine is None else mine.shape[1:])
        shape = next(s_ for s_ in shapes if s_ is not None)
        if mine is None:
            mine = np.empty((0,) + tuple(shape), dtype=np.float32)
        if not gather:
            return lo, mine
        # gather on rank 0 (ONLY rank 0 returns the complete stack; the others return their own range): equal-sized
        # (padded) blocks through the backend's device, a bounded number of frames at a time so that a 17 GB stack never
        # needs a second full copy on one GPU
        cnt = max(self.frame_range(n, r, world)[1] - self.frame_range(n, r, world)[0] for r in range(world))
        dev = self.device if dist.get_backend() == "nccl" else "cpu"
        out = np.empty((n,) + tuple(shape), dtype=np.float32) if rank == 0 else None
        step = max(1, (256 << 20) // (int(np.prod(shape)) * 4))
        for s in range(0, cnt, step):
            m = min(step, cnt - s)
            blk = torch.zeros((m,) + tuple(shape), dtype=torch.float32, device=dev)
            if s < len(mine):
                k = min(m, len(mine) - s)
                blk[:k] = torch.from_numpy(mine[s:s + k]).to(dev)
            bufs = [torch.empty_like(blk) for _ in range(world)] if rank == 0 else None
            dist.gather(blk, bufs, dst=0)
            if rank == 0:
                for r in range(world):
                    rlo, rhi = self.frame_range(n, r, world)
                    k = min(m, (rhi - rlo) - s)
                    if k > 0:
                        out[rlo + s: rlo + s + k] = bufs[r][:k].cpu().numpy()
        return out if rank == 0 else mine

    def predict(self, image_data: np.ndarray, return_image: bool = False, **kwargs: int):
        if kwargs.pop("distributed", False):
            if return_image:
                raise NotImplementedError("return_image is not available with distributed=True")
            return self.predict_distributed(image_data, **kwargs)
        kwargs.pop("gather", None)
        with _host_threads():
            image_data = self.preprocess(image_data, kwargs.get("norm", True), device_norm=not return_image)
            n, _, w, h = image_data.shape
            num_batches = kwargs.get("num_batches")
            if num_batches is None:
                num_batches = len(image_data) if (w >= 256 or h >= 256) else 10
            segmented = self.batch_predict(image_data, (n, w, h, self.nb_classes), num_batches,
                                           kwargs.get("_on_chunk"))
        if return_image:
            return image_data.permute(0, 2, 3, 1).numpy(), segmented.numpy()
        return segmented.numpy()

    def run(self, image_data: np.ndarray, compute_coords=True, **kwargs: int):
        """Prediction (+ blob centres).  With ``compute_coords`` the Locator kernels run on each chunk's
        probabilities while they are still on the device (predictor.py:262-298 runs a host loop afterwards)."""
        start_time = time.time()
        if not compute_coords:
            return self.predict(image_data, **kwargs)       # the reference prints nothing on this branch
        if self.refine:
            raise NotImplementedError("peak refinement (per-atom scipy.optimize Gaussian fits) is outside the "
                                      "MI355X hot path of this build")
        thresh = kwargs.get("thresh", self.thresh)
        dist_edge = Locator(thresh).dist_edge               # Locator's default, as the reference uses it
        coordinates = {}

        def on_chunk(start, prob):
            for i, v in locate_device(prob, thresh, dist_edge).items():
                coordinates[start + i] = v
        kw = {k: v for k, v in kwargs.items() if k != "thresh"}
        decoded = self.predict(image_data, _on_chunk=on_chunk, **kw)
        if kwargs.get("distributed", False) and self._dist_world()[1] > 1:
            # frame indices seen by on_chunk are local to this rank's range: shift, then merge on rank 0
            import torch.distributed as dist
            rank, world = self._dist_world()
            n = 1 if image_data.ndim == 2 else len(image_data)
            lo = self.frame_range(n, rank, world)[0]
            coordinates = {lo + i: v for i, v in coordinates.items()}
            if kwargs.get("gather", True):
                parts = [None] * world
                dist.all_gather_object(parts, coordinates)
                if rank == 0:
                    coordinates = {k: v for part in parts for k, v in part.items()}
        coordinates = {i: coordinates[i] for i in sorted(coordinates)}
        if self.verbose:
            word = " image was " if decoded.shape[0] == 1 else " images were "
            print("\n" + str(decoded.shape[0]) + word + "decoded in approximately "
                  + str(np.around(time.time() - start_time, decimals=4)) + " seconds")
        return decoded, coordinates
