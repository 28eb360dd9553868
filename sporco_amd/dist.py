"""Image-axis sharding over the GPUs of a node (one process per GPU).

For a fixed rho every update of the ConvBPDN iteration is independent per
image; the only coupling between images is through the
scalars that drive the rho schedule and the stopping test.  A rank therefore
owns a contiguous block of images (its own ``S[..., n0:n1]``) and the ranks
exchange nothing but one all-reduce of the 16 per-iteration sums -- over RCCL
(``backend='nccl'`` on ROCm) when the tensors live on the GPU, over gloo in the
CPU test-suite.  This mirrors the reference's per-image split in
``sporco/dictlrn/prlcnscdl.py:241,508`` (there: multiprocessing + shared memory).

Dictionary learning shards the same way: the sparse coding step as
above, and in the dictionary step the gradient -- a sum over images of a dictionary-sized
spectrum, 17 MB at config 5 -- is all-reduced in place on the device
(:meth:`TorchReducer.all_reduce_array`); the constraint projection is replicated.

``torch`` is used here for the process group only; it is not imported by the
single-GPU path.
"""

import ctypes


class _DeviceView(object):
    """Flat float view of library-owned device memory for ``torch.as_tensor`` (CUDA array
    interface v2; ROCm builds of torch implement the same protocol)."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {'shape': (int(count),), 'typestr': typestr,
                                         'data': (int(ptr), False), 'version': 2}


class TorchReducer(object):
    """Sums per-iteration scalars across the ranks of a torch process group."""

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        backend = dist.get_backend(group)
        self.on_gpu = backend == 'nccl'
        if self.on_gpu:
            dev = torch.device('cuda', torch.cuda.current_device()) if device is None else device
            self.buf = torch.zeros(16, dtype=torch.float64, device=dev)
        else:
            self.buf = torch.zeros(16, dtype=torch.float64)

    def stream_handle(self):
        """hipStream_t of torch's current stream (share it with the solver so
        the all-reduce is ordered after the kernels that produce the sums)."""
        return self.torch.cuda.current_stream().cuda_stream if self.on_gpu else None

    def admm_iter(self, solver, params):
        if self.on_gpu:
            solver.admm_iter_dev(params, self.buf.data_ptr())
            # The sums are produced on the solver's stream and consumed by RCCL on a stream
            # of its own: make the hand-over explicit instead of relying on the implicit
            # ordering of the legacy default stream (one host sync of ~10 us per iteration).
            solver.sync()
            self.dist.all_reduce(self.buf, group=self.group)
            return self.buf.cpu().tolist()
        return self.sum(solver.admm_iter(params))

    def device_sum_hook(self, solver):
        """Hook for the device-driven solve (``Solver.admm_run``): called once per iteration
        with the device address of the 16 sums, it all-reduces them in place.

        RCCL (``nccl``): no host round trip.  The collective is issued with the SOLVER's stream
        as torch's current stream (``torch.cuda.ExternalStream`` over the handle's
        ``hipStream_t``), which is how ProcessGroupNCCL orders a collective: it records an
        event on the current stream, makes its own communication stream wait for it, and --
        for a blocking call -- makes the current stream wait for the collective's end event.
        So the all-reduce runs behind the kernels that produced the sums and the control
        kernel the library enqueues next runs behind the all-reduce, whichever stream the
        handle was created with; nothing relies on the legacy default stream.

        gloo with the library on a real GPU (two ranks sharing one device: a CORRECTNESS
        configuration, e.g. ``SPORCO_AMD_BENCH_BACKEND=gloo`` -- not a performance path): the 16
        doubles are staged through the host around a CPU all-reduce, i.e. one full stream
        synchronisation, one device-to-host and one host-to-device copy per iteration, which
        also defeats the look-ahead of ``admm_run``.

        gloo on the CPU simulator: "device" memory is host memory."""
        from . import _lib
        torch = self.torch
        views = {}
        # per-call cost of the collective, for attribution of a scaling shortfall (bench.py):
        # device time between two events around the all-reduce (RCCL), or host time of the CPU
        # all-reduce without the stream synchronisation that precedes it (gloo)
        self.hook_calls = 0
        self._hook_events, self._hook_host_s = [], []
        if self.on_gpu:
            try:
                torch.as_tensor(_DeviceView(solver.device_ptr(_lib.VAR_Y), 1, '<f4'),
                                device=self.buf.device)
                ext = torch.cuda.ExternalStream(solver.stream_handle(), device=self.buf.device)
            except (TypeError, RuntimeError, ValueError, AttributeError):
                return None

            def hook(ptr):
                t = views.get(ptr)
                if t is None:
                    t = views[ptr] = torch.as_tensor(_DeviceView(ptr, 16, '<f8'),
                                                     device=self.buf.device)
                with torch.cuda.stream(ext):
                    if self.hook_calls < 256:      # (timed for the first 256 calls)
                        e0 = torch.cuda.Event(enable_timing=True)
                        e1 = torch.cuda.Event(enable_timing=True)
                        e0.record()
                        self.dist.all_reduce(t, group=self.group)
                        e1.record()
                        self._hook_events.append((e0, e1))
                    else:
                        self.dist.all_reduce(t, group=self.group)
                self.hook_calls += 1
            return hook
        if 'hostsim' not in str(_lib.library_path()):
            import time
            import numpy as np
            stage = np.zeros(16, dtype=np.float64)
            tstage = torch.from_numpy(stage)

            def hook(ptr):      # real GPU, CPU collective: host staging
                solver.sync()
                _lib.check(_lib.lib().sporco_amd_dev_download(_lib._ptr(stage), ctypes.c_void_p(ptr),
                                                              stage.nbytes))
                t0 = time.perf_counter()
                self.dist.all_reduce(tstage, group=self.group)
                if self.hook_calls < 256:      # (timed for the first 256 calls, as the RCCL branch)
                    self._hook_host_s.append(time.perf_counter() - t0)
                self.hook_calls += 1
                _lib.check(_lib.lib().sporco_amd_dev_upload(ctypes.c_void_p(ptr), _lib._ptr(stage),
                                                            stage.nbytes))
            return hook

        def hook(ptr):      # CPU simulator: "device" memory is host memory
            import numpy as np
            t = views.get(ptr)
            if t is None:
                a = np.ctypeslib.as_array((ctypes.c_double * 16).from_address(ptr))
                t = views[ptr] = torch.from_numpy(a)
            import time
            t0 = time.perf_counter()
            self.dist.all_reduce(t, group=self.group)
            if self.hook_calls < 256:
                self._hook_host_s.append(time.perf_counter() - t0)
            self.hook_calls += 1
        return hook

    def hook_cost_ms(self):
        """(mean, max, calls) of the all-reduce inside the device-driven loop's hook, in
        milliseconds, over the timed calls since the hook was made; None when it never ran."""
        if getattr(self, '_hook_events', None):
            self.torch.cuda.synchronize()
            ms = [a.elapsed_time(b) for a, b in self._hook_events]
        elif getattr(self, '_hook_host_s', None):
            ms = [1e3 * t for t in self._hook_host_s]
        else:
            return None
        return (sum(ms) / len(ms), max(ms), len(ms))

    def all_reduce_array(self, solver, var):
        """Sum state array ``var`` of ``solver`` over the ranks, in place, in device memory."""
        ptr = solver.device_ptr(var)
        count, real = solver.device_reals(var)
        solver.sync()                      # produced on the solver's stream
        if self.on_gpu:
            typestr = '<f4' if real == self.torch.float32 else '<f8'
            try:
                t = self.torch.as_tensor(_DeviceView(ptr, count, typestr),
                                         device=self.buf.device)
                self.array_transport = 'device'
            except (TypeError, RuntimeError, ValueError) as exc:
                # a torch build without the array-interface import: stage through the host
                # (correct, slow -- said once, loudly)
                if getattr(self, 'array_transport', None) != 'host-staged':
                    import warnings
                    warnings.warn("sporco_amd.dist: torch cannot alias library device memory "
                                  "(%s); all-reducing arrays through host staging" % exc)
                self.array_transport = 'host-staged'
                a = solver.download(var)
                t = self.torch.from_numpy(a).to(self.buf.device)
                self.dist.all_reduce(t, group=self.group)
                solver.upload(var, t.cpu().numpy())
                return
            self.dist.all_reduce(t, group=self.group)
            self.torch.cuda.current_stream().synchronize()   # consumed on the solver's stream
        else:
            import numpy as np
            ct = ctypes.c_float if real == self.torch.float32 else ctypes.c_double
            a = np.ctypeslib.as_array((ct * count).from_address(ptr))
            self.dist.all_reduce(self.torch.from_numpy(a), group=self.group)

    def all_reduce_ptr(self, solver, ptr, count, f32, prescale=1.0):
        """Sum ``count`` reals at device address ``ptr`` of ``solver`` over the ranks, in place,
        each rank's contribution first multiplied by ``prescale`` (a local mean over N_local
        images becomes the global mean with prescale = N_local / N_total)."""
        solver.sync()
        typestr = '<f4' if f32 else '<f8'
        if self.on_gpu:
            t = self.torch.as_tensor(_DeviceView(ptr, count, typestr), device=self.buf.device)
            if prescale != 1.0:
                t.mul_(prescale)
            self.dist.all_reduce(t, group=self.group)
            self.torch.cuda.current_stream().synchronize()
            return
        import numpy as np
        ct = ctypes.c_float if f32 else ctypes.c_double
        a = np.ctypeslib.as_array((ct * count).from_address(ptr))
        if prescale != 1.0:
            a *= a.dtype.type(prescale)
        self.dist.all_reduce(self.torch.from_numpy(a), group=self.group)

    def sum_slots(self, values, slots):
        """``values`` with the entries at ``slots`` summed over the ranks (the others are
        replicated quantities and stay as they are)."""
        red = self.sum([values[i] for i in slots])
        out = list(values)
        for i, v in zip(slots, red):
            out[i] = v
        return out

    def sum(self, values):
        t = self.torch.tensor(list(values), dtype=self.torch.float64)
        if self.on_gpu:
            t = t.to(self.buf.device)
        self.dist.all_reduce(t, group=self.group)
        return t.cpu().tolist()

    def max(self, value):
        t = self.torch.tensor([float(value)], dtype=self.torch.float64)
        if self.on_gpu:
            t = t.to(self.buf.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.cpu()[0])


class NativeReducer(object):
    """Image shards over RCCL INSIDE the library (``sporco_amd_comm_*``, csc_comm.hip): the
    device-driven solve enqueues the all-reduce of its 16 per-iteration doubles on the solver's
    stream itself -- no Python callback per iteration, and ``torch`` is not needed (the
    reference-side counterpart is the per-image split of sporco/dictlrn/prlcnscdl.py:241,508).

    Every rank builds one from the same 128-byte id: rank 0 calls :meth:`unique_id` and hands the
    bytes to the others by whatever the host program has (an MPI broadcast, a file, torch's
    store -- :meth:`from_torch` does the latter when torch.distributed is initialised anyway; with
    a single rank nothing has to be exchanged).  Same interface as :class:`TorchReducer`."""

    COMM_SUM, COMM_MAX = 0, 2

    def __init__(self, rank, world_size, unique_id, device=0):
        from . import _lib
        self._lib = _lib.lib()
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of NativeReducer.unique_id()")
        idbuf = (ctypes.c_char * 128).from_buffer_copy(bytes(unique_id))
        h = ctypes.c_void_p()
        _lib.check(self._lib.sporco_amd_comm_create(idbuf, int(rank), int(world_size), int(device),
                                                    ctypes.byref(h)))
        self._h = h
        self.rank, self.world_size, self.device = int(rank), int(world_size), int(device)
        self.on_gpu = True

    @staticmethod
    def unique_id():
        """128 bytes identifying a new communicator (ncclGetUniqueId); call on one rank."""
        from . import _lib
        buf = (ctypes.c_char * 128)()
        _lib.check(_lib.lib().sporco_amd_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def from_torch(cls, group=None, device=None):
        """Bootstrap through an initialised torch.distributed process group: rank 0's id is
        broadcast as an object; the collectives themselves do not go through torch."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        return cls(rank, world, box[0], device=device)

    def close(self):
        if getattr(self, '_h', None):
            # solvers that still hold the raw communicator pointer must not use it after this
            for ref in getattr(self, '_attached', ()):
                solver = ref()
                if solver is not None:
                    try:
                        solver.set_comm(None)
                    except Exception:
                        pass
            self._attached = []
            self._lib.sporco_amd_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stream_handle(self):
        return None        # (the solver keeps its own stream; the collective is enqueued on it)

    # -- per-iteration scalars ----------------------------------------------------------------
    def device_sum_hook(self, solver):
        """The device-driven solve needs no hook: the library all-reduces on the solver's
        stream once the communicator is attached.  Returns the marker the caller passes on."""
        solver.set_comm(self._h.value)
        import weakref
        if not hasattr(self, '_attached'):
            self._attached = []
        if not any(r() is solver for r in self._attached):
            self._attached.append(weakref.ref(solver))
        return False       # (not None: "sharded, and the library does the reduction itself")

    def admm_iter(self, solver, params):
        return self.sum(solver.admm_iter(params))

    def _host(self, values, op):
        from . import _lib
        vals = list(values)
        out = []
        for i in range(0, len(vals), 64):
            chunk = vals[i:i + 64]
            buf = (ctypes.c_double * len(chunk))(*[float(v) for v in chunk])
            _lib.check(self._lib.sporco_amd_comm_allreduce_host(self._h, buf, len(chunk), op))
            out.extend(buf)
        return out

    def sum(self, values):
        return self._host(values, self.COMM_SUM)

    def max(self, value):
        return self._host([value], self.COMM_MAX)[0]

    def sum_slots(self, values, slots):
        red = self.sum([values[i] for i in slots])
        out = list(values)
        for i, v in zip(slots, red):
            out[i] = v
        return out

    # -- arrays in device memory --------------------------------------------------------------
    def all_reduce_ptr(self, solver, ptr, count, f32, prescale=1.0):
        from . import _lib
        solver.sync()      # (the prescale runs on the null stream: the producer of `ptr` first)
        if prescale != 1.0:
            _lib.check(self._lib.sporco_amd_dev_axpby(_lib.F32 if f32 else _lib.F64, int(count),
                                                      float(prescale), ctypes.c_void_p(ptr), 0.0,
                                                      ctypes.c_void_p(ptr), ctypes.c_void_p(ptr)))
        solver.sync()
        _lib.check(self._lib.sporco_amd_comm_allreduce(self._h, ctypes.c_void_p(ptr), int(count),
                                                       _lib.F32 if f32 else _lib.F64, self.COMM_SUM,
                                                       ctypes.c_void_p(solver.stream_handle())))
        solver.sync()

    def all_reduce_array(self, solver, var):
        import numpy as np
        ptr = solver.device_ptr(var)
        shape, dt = solver.var_shape_dtype(var)
        from . import _lib
        kdev = solver.query(_lib.QUERY_DEVICE_FILTERS)
        n = int(np.prod(shape[:-1])) * (kdev if shape[-1] == solver.dims[4] else shape[-1])
        if np.dtype(dt).kind == 'c':
            n *= 2
        f32 = np.dtype(dt) in (np.dtype(np.float32), np.dtype(np.complex64))
        self.all_reduce_ptr(solver, ptr, n, f32)

    def hook_cost_ms(self):
        return None


class ReducingSolver(object):
    """Stand-in for a :class:`sporco_amd._lib.Solver` holding one rank's images: every call
    that returns sums over the coefficient arrays (objective terms, residuals, the inner
    products of the step-size and backtracking policies) returns them summed over the ranks,
    everything else goes straight to the device handle (``_raw``).  Used by the FISTA sparse
    coding step, whose host logic asks the device for such sums in many places."""

    _SUMS = ('pgm_iter', 'pgm_grad', 'pgm_eval', 'pgm_prox_step', 'pair_stats', 'masked_grad')

    def __init__(self, raw, reducer):
        self.__dict__['_raw'] = raw
        self.__dict__['_reducer'] = reducer

    def __getattr__(self, name):
        attr = getattr(self._raw, name)
        red = self._reducer
        if name in self._SUMS:
            return lambda *a, **k: red.sum(attr(*a, **k))
        if name == 'asum':
            return lambda *a, **k: red.sum([attr(*a, **k)])[0]
        if name == 'dhs_absmax':
            return lambda *a, **k: red.max(attr(*a, **k))
        return attr

    def __setattr__(self, name, value):
        setattr(self._raw, name, value)


def shard_bounds(n, rank, world_size):
    """[lo, hi) of the contiguous block of ``n`` images owned by ``rank``: blocks differ by at most
    one image, the first ``n % world_size`` ranks hold the larger ones (the per-image split of
    the reference's parallel solvers has no divisibility condition either, prlcnscdl.py:241)."""
    if n < world_size:
        raise ValueError("%d images cannot be spread over %d ranks (every rank needs one)" %
                         (n, world_size))
    per, extra = divmod(n, world_size)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


def shard_images(S, rank, world_size, axis=-1):
    """Contiguous block of the image axis owned by ``rank`` (see :func:`shard_bounds`)."""
    import numpy as np
    lo, hi = shard_bounds(S.shape[axis], rank, world_size)
    idx = [slice(None)] * S.ndim
    idx[axis] = slice(lo, hi)
    return np.ascontiguousarray(S[tuple(idx)])


def global_count(reducer, local):
    """Sum over the ranks of a per-rank count (images, blocks, elements): what the residual
    tolerances and default step sizes of a sharded solver refer to.  A collective: every rank
    calls it at the same point (the constructors do)."""
    if reducer is None:
        return int(local)
    return int(round(reducer.sum([float(local)])[0]))
