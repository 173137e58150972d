"""ctypes binding of the engine's C-ABI (include/tpamd.h -> csrc/libtpamd.so).

Plumbing only: device memory, streams and torch.distributed come from PyTorch;
all computation happens in the hand-written HIP kernels behind the C-ABI. There
is NO CPU fallback: if the shared library or a HIP device is missing, every
entry point raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_SO = os.environ.get("TPAMD_LIBRARY") or os.path.join(_CSRC, "libtpamd.so")   # override: A/B builds
_SOURCES = ["tpamd_capi.hip", "tpamd_sweep_inst.hip", "tpamd_launch.h", "tpamd_kernels.h", "tpamd_device.h",
            "tpamd_sweep_joint.h", "tpamd_planner_set.h"]
_HEADER = os.path.join(os.path.dirname(_HERE), "include", "tpamd.h")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared",
               "-std=c++17"]

PATH_STATUS = {
    0: "ok", 2: "infeasible_bounds", 3: "s_range", 4: "sd_start_negative",
    5: "lower_ge_upper", 6: "too_few_samples", 7: "no_connection", 8: "nan_sd2",
    9: "nonzero_end", 10: "crit_index_zero",
}

KERNEL_SWEEP = 4


class TpamdError(RuntimeError):
    pass


# (joint count, extra rows) instances of the specialised sweep kernel: one translation unit each
# (csrc/tpamd_launch.h TPAMD_SWEEP_INSTANCES must list the same pairs)
SWEEP_INSTANCES = [(3, 0), (4, 0), (5, 0), (6, 0), (7, 0), (8, 0), (14, 0), (6, 2), (7, 2)]


def _compile(target, extra_flags, force, verbose):
    """hipcc -c every translation unit (in parallel: the sweep instances take 10-18 s each), then
    link. Objects are kept per target under build/ so that a rebuild recompiles everything only
    when a source changed."""
    deps = [os.path.join(_CSRC, s) for s in _SOURCES] + [_HEADER]
    stale = force or not os.path.exists(target) or any(
        os.path.getmtime(d) > os.path.getmtime(target) for d in deps)
    if not stale:
        return target
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(os.path.dirname(_HERE), "build",
                          "obj_" + os.path.splitext(os.path.basename(target))[0])
    os.makedirs(objdir, exist_ok=True)
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + list(extra_flags)
    units = [(os.path.join(objdir, "capi.o"), ["tpamd_capi.hip"])]
    for d, e in SWEEP_INSTANCES:
        units.append((os.path.join(objdir, "sweep_%d_%d.o" % (d, e)),
                      ["-DTPAMD_INST_D=%d" % d, "-DTPAMD_INST_E=%d" % e, "tpamd_sweep_inst.hip"]))

    def one(unit):
        obj, args = unit
        cmd = ["hipcc"] + flags + ["-c", "-o", obj] + args[:-1] + [os.path.join(_CSRC, args[-1])]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=_CSRC)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(one, units))
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=_CSRC)
    return target


def build_library(force=False, verbose=False, extra_flags=(), output=None):
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    global _SO
    if output is not None:
        _SO = output
    return _compile(_SO, extra_flags, force, verbose)


MULTI_SO = os.path.join(_CSRC, "libtpamd_multi.so")


def build_multi_library(force=False, verbose=False):
    """libtpamd_multi.so (include/tpamd_multi.h): several devices from one process, on top of
    libtpamd.so and RCCL. Host code only."""
    src = os.path.join(_CSRC, "tpamd_multi.cc")
    deps = [src, _HEADER, os.path.join(os.path.dirname(_HEADER), "tpamd_multi.h"), _SO]
    if force or not os.path.exists(MULTI_SO) or any(
            os.path.getmtime(d) > os.path.getmtime(MULTI_SO) for d in deps if os.path.exists(d)):
        cmd = ["hipcc", "-O2", "-fPIC", "-shared", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", src,
               "-o", MULTI_SO, "-L" + _CSRC, "-ltpamd", "-L/opt/rocm/lib", "-lrccl", "-lpthread",
               "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=_CSRC)
    return MULTI_SO


DIAG_SO = os.path.join(_CSRC, "libtpamd_diag.so")


def build_diagnostic_library(force=False, verbose=False):
    """The -DTPAMD_DIAG variant: in-kernel cycle counters and the literal cross-checks of the
    sweep kernel's shortcuts (tests/test_gpu_parity.py, tools/gpu_diag.py). Not the product."""
    return _compile(DIAG_SO, ("-DTPAMD_DIAG",), force, verbose)


class _JointBatch(C.Structure):
    _fields_ = [("num_paths", C.c_int32), ("num_dofs", C.c_int32), ("num_samples", C.c_int32),
                ("num_points", C.c_int32), ("max_solver_loops", C.c_int32),
                ("reserved", C.c_int32), ("constraint_safety", C.c_double)]


class _JointInputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "knots", "control_points", "max_velocity", "max_acceleration", "path_start", "delta",
        "sd_start", "sdd_start", "time_start", "num_samples_per_path")]


class _PathOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "time", "s", "sd", "sdd", "q", "qd", "qdd", "last_extremal_index",
        "max_time_increment", "status", "sd2")]


class _RowsBatch(C.Structure):
    _fields_ = [("num_paths", C.c_int32), ("num_samples", C.c_int32), ("num_rows", C.c_int32),
                ("max_solver_loops", C.c_int32)]


class _RowsInputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "a", "b", "lower", "upper", "s_start", "s_end", "sd_start", "sdd_start", "time_start")]


class _CartesianBatch(C.Structure):
    _fields_ = [("num_paths", C.c_int32), ("num_dofs", C.c_int32), ("num_samples", C.c_int32),
                ("max_solver_loops", C.c_int32), ("constraint_safety", C.c_double)]


_CARTESIAN_INPUT_KEYS = ("ik_positions", "jacobians", "max_velocity", "max_acceleration",
                         "max_translational_velocity", "max_rotational_velocity", "path_start",
                         "delta", "sd_start", "sdd_start", "time_start")


class _CartesianInputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _CARTESIAN_INPUT_KEYS]


class _ResampleArgs(C.Structure):
    _fields_ = ([("num_paths", C.c_int32), ("num_samples", C.c_int32), ("num_dofs", C.c_int32),
                 ("max_out", C.c_int32)] +
                [(n, C.c_void_p) for n in ("time", "s", "sd", "sdd", "q", "qd", "qdd",
                                           "max_acceleration", "start_sec")] +
                [("time_step", C.c_double), ("status", C.c_void_p)] +
                [(n, C.c_void_p) for n in ("out_time", "out_s", "out_sd", "out_sdd", "out_q",
                                           "out_qd", "out_qdd", "count")])


_LIB = None

# every symbol include/tpamd.h declares
ABI_SYMBOLS = [
    "tpamd_engine_create", "tpamd_engine_destroy", "tpamd_version", "tpamd_error_string",
    "tpamd_device_count", "tpamd_shard_bounds", "tpamd_shard_bounds_balanced",
    "tpamd_engine_reserve", "tpamd_engine_set_pipelining", "tpamd_engine_fence",
    "tpamd_engine_workspace_bytes",
    "tpamd_time_joint_paths_device",
    "tpamd_time_joint_paths_host", "tpamd_sample_joint_paths_host",
    "tpamd_time_joint_groups_device", "tpamd_time_joint_groups_host",
    "tpamd_optimize_rows_device", "tpamd_optimize_rows_host",
    "tpamd_time_cartesian_paths_device", "tpamd_time_cartesian_paths_host",
    "tpamd_sample_pose_splines_device", "tpamd_sample_pose_splines_host",
    "tpamd_plan_joint_windows_host",
    "tpamd_planner_set_create", "tpamd_planner_set_destroy", "tpamd_planner_set_upload_paths",
    "tpamd_planner_set_reset", "tpamd_planner_set_plan", "tpamd_planner_set_download_trajectory",
    "tpamd_planner_set_last_plan_bytes", "tpamd_planner_set_device_bytes",
    "tpamd_find_max_sd2_host", "tpamd_query_device", "tpamd_resample_uniform_device",
    "tpamd_resample_uniform_host", "tpamd_resample_skip_device", "tpamd_resample_skip_host",
    "tpamd_debug_copy_boundary", "tpamd_debug_keep_boundary", "tpamd_debug_copy_diag", "tpamd_debug_kernel_vgprs",
    "tpamd_rebuild_time_device", "tpamd_profile_reset", "tpamd_profile_enable",
    "tpamd_profile_mean_ms", "tpamd_profile_kernel_name", "tpamd_profile_num_kernels",
]


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_SO):
        raise TpamdError(
            "libtpamd.so is not built (%s). Run __graft_entry__.build(); there is no CPU "
            "fallback for the engine." % _SO)
    # PyTorch ships its own HIP runtime; two runtimes in one process do not share devices.
    # Import torch first so that libtpamd.so binds to the runtime torch uses, whatever the
    # order the caller imports things in.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(_SO)
    vp, i = C.c_void_p, C.c_int
    L.tpamd_engine_create.restype = i
    L.tpamd_engine_create.argtypes = [i, C.POINTER(vp)]
    L.tpamd_engine_destroy.argtypes = [vp]
    L.tpamd_version.restype = i
    L.tpamd_error_string.restype = C.c_char_p
    L.tpamd_error_string.argtypes = [i]
    L.tpamd_device_count.restype = i
    L.tpamd_shard_bounds.restype = None
    L.tpamd_shard_bounds.argtypes = [i, i, vp]
    L.tpamd_shard_bounds_balanced.restype = None
    L.tpamd_shard_bounds_balanced.argtypes = [i, vp, i, vp]
    L.tpamd_engine_reserve.restype = i
    L.tpamd_engine_reserve.argtypes = [vp, i, i, i]
    L.tpamd_engine_set_pipelining.restype = i
    L.tpamd_engine_set_pipelining.argtypes = [vp, i]
    L.tpamd_engine_fence.restype = i
    L.tpamd_engine_fence.argtypes = [vp, vp]
    L.tpamd_engine_workspace_bytes.restype = C.c_size_t
    L.tpamd_engine_workspace_bytes.argtypes = [vp]
    L.tpamd_time_joint_paths_device.restype = i
    L.tpamd_time_joint_paths_device.argtypes = [vp, C.POINTER(_JointBatch),
                                                C.POINTER(_JointInputs),
                                                C.POINTER(_PathOutputs), vp]
    L.tpamd_time_joint_paths_host.restype = i
    L.tpamd_time_joint_paths_host.argtypes = [vp, C.POINTER(_JointBatch),
                                              C.POINTER(_JointInputs), C.POINTER(_PathOutputs)]
    L.tpamd_time_joint_groups_device.restype = i
    L.tpamd_time_joint_groups_device.argtypes = [vp, i, C.POINTER(_JointBatch), C.POINTER(_JointInputs),
                                                 C.POINTER(_PathOutputs), vp]
    L.tpamd_time_joint_groups_host.restype = i
    L.tpamd_time_joint_groups_host.argtypes = [vp, i, C.POINTER(_JointBatch), C.POINTER(_JointInputs),
                                               C.POINTER(_PathOutputs)]
    L.tpamd_sample_joint_paths_host.restype = i
    L.tpamd_sample_joint_paths_host.argtypes = [vp, i, i, i, i] + [vp] * 7
    L.tpamd_optimize_rows_device.restype = i
    L.tpamd_optimize_rows_device.argtypes = [vp, C.POINTER(_RowsBatch), C.POINTER(_RowsInputs),
                                             C.POINTER(_PathOutputs), vp]
    L.tpamd_optimize_rows_host.restype = i
    L.tpamd_optimize_rows_host.argtypes = [vp, C.POINTER(_RowsBatch), C.POINTER(_RowsInputs),
                                           C.POINTER(_PathOutputs)]
    L.tpamd_time_cartesian_paths_device.restype = i
    L.tpamd_time_cartesian_paths_device.argtypes = [vp, C.POINTER(_CartesianBatch),
                                                    C.POINTER(_CartesianInputs),
                                                    C.POINTER(_PathOutputs), vp]
    L.tpamd_time_cartesian_paths_host.restype = i
    L.tpamd_time_cartesian_paths_host.argtypes = [vp, C.POINTER(_CartesianBatch),
                                                  C.POINTER(_CartesianInputs),
                                                  C.POINTER(_PathOutputs)]
    L.tpamd_sample_pose_splines_host.restype = i
    L.tpamd_sample_pose_splines_host.argtypes = [vp, i, i, i] + [vp] * 6
    L.tpamd_sample_pose_splines_device.restype = i
    L.tpamd_sample_pose_splines_device.argtypes = [vp, i, i, i] + [vp] * 6 + [vp]
    L.tpamd_find_max_sd2_host.restype = i
    L.tpamd_find_max_sd2_host.argtypes = [vp, i, i] + [vp] * 7
    L.tpamd_query_device.restype = i
    L.tpamd_query_device.argtypes = [vp, i, i, i] + [vp] * 10 + [vp]
    L.tpamd_resample_uniform_device.restype = i
    L.tpamd_resample_uniform_device.argtypes = [vp, C.POINTER(_ResampleArgs), vp]
    L.tpamd_resample_uniform_host.restype = i
    L.tpamd_resample_uniform_host.argtypes = [vp, C.POINTER(_ResampleArgs)]
    L.tpamd_resample_skip_device.restype = i
    L.tpamd_resample_skip_device.argtypes = [vp, C.POINTER(_ResampleArgs), vp]
    L.tpamd_resample_skip_host.restype = i
    L.tpamd_resample_skip_host.argtypes = [vp, C.POINTER(_ResampleArgs)]
    L.tpamd_debug_copy_boundary.restype = i
    L.tpamd_debug_copy_boundary.argtypes = [vp, i, i] + [vp] * 6
    L.tpamd_debug_keep_boundary.argtypes = [vp, i]
    L.tpamd_debug_copy_diag.restype = i
    L.tpamd_debug_copy_diag.argtypes = [vp, i, vp]
    L.tpamd_debug_kernel_vgprs.restype = i
    L.tpamd_debug_kernel_vgprs.argtypes = [vp, i]
    L.tpamd_rebuild_time_device.restype = i
    L.tpamd_rebuild_time_device.argtypes = [vp, i, i, i, C.c_size_t, vp, vp, vp, vp, vp, vp]
    L.tpamd_profile_reset.argtypes = [vp]
    L.tpamd_profile_enable.argtypes = [vp, i]
    L.tpamd_profile_mean_ms.restype = C.c_double
    L.tpamd_profile_mean_ms.argtypes = [vp, i, C.POINTER(i)]
    L.tpamd_profile_kernel_name.restype = C.c_char_p
    L.tpamd_profile_kernel_name.argtypes = [i]
    L.tpamd_profile_num_kernels.restype = i
    _LIB = L
    return L


def _check(rc, what):
    if rc != 0:
        raise TpamdError("%s failed: %d (%s)" % (what, rc,
                                                load_library().tpamd_error_string(rc).decode()))


def _ptr(t):
    """Device/host pointer of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        assert t.flags["C_CONTIGUOUS"]
        return t.ctypes.data
    assert t.is_contiguous()
    return t.data_ptr()


class Engine:
    """One engine per GPU (tpamd_engine_create/destroy)."""

    def __init__(self, device=0):
        self._lib = load_library()
        h = C.c_void_p()
        _check(self._lib.tpamd_engine_create(int(device), C.byref(h)), "tpamd_engine_create")
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tpamd_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reserve(self, num_paths, num_samples, num_rows):
        _check(self._lib.tpamd_engine_reserve(self._h, num_paths, num_samples, num_rows),
               "tpamd_engine_reserve")

    def set_pipelining(self, mode=1):
        """0 off; 1: the sampling/LP kernel of a joint solve overlaps the previous solve's sweep;
        2: sweeps of consecutive solves overlap as well, outputs ordered by the next call or
        fence() (include/tpamd.h tpamd_engine_set_pipelining: inputs must be ready at call time)."""
        _check(self._lib.tpamd_engine_set_pipelining(self._h, int(mode)),
               "tpamd_engine_set_pipelining")

    def fence(self, stream=None):
        """Order `stream` (default: torch's current stream) behind every solve issued so far."""
        _check(self._lib.tpamd_engine_fence(self._h, _stream_ptr(stream)), "tpamd_engine_fence")

    @property
    def workspace_bytes(self):
        return self._lib.tpamd_engine_workspace_bytes(self._h)

    # ------------------------------------------------------------ joint paths
    def time_joint_paths(self, inputs, outputs, num_samples, safety=0.8, max_solver_loops=0,
                         stream=None, host=False):
        """inputs: dict with knots [B][P+3], control_points [B][P][D], max_velocity,
        max_acceleration [B][D], path_start, delta, sd_start, time_start [B] (+ optional
        sdd_start). outputs: dict with time, s, sd, sdd [B][N], status [B] int32 and optional
        q, qd, qdd [B][N][D], last_extremal_index [B] int32, max_time_increment [B].
        Torch CUDA tensors (host=False) or numpy arrays (host=True)."""
        cp = inputs["control_points"]
        B, P, D = cp.shape
        bt = _JointBatch(B, D, int(num_samples), P, int(max_solver_loops), 0, float(safety))
        ji = _JointInputs(*[_ptr(inputs.get(k)) for k in (
            "knots", "control_points", "max_velocity", "max_acceleration", "path_start", "delta",
            "sd_start", "sdd_start", "time_start", "num_samples_per_path")])
        po = _PathOutputs(*[_ptr(outputs.get(k)) for k in (
            "time", "s", "sd", "sdd", "q", "qd", "qdd", "last_extremal_index",
            "max_time_increment", "status", "sd2")])
        if host:
            _check(self._lib.tpamd_time_joint_paths_host(self._h, C.byref(bt), C.byref(ji),
                                                         C.byref(po)),
                   "tpamd_time_joint_paths_host")
        else:
            _check(self._lib.tpamd_time_joint_paths_device(self._h, C.byref(bt), C.byref(ji),
                                                           C.byref(po), _stream_ptr(stream)),
                   "tpamd_time_joint_paths_device")

    def time_joint_groups(self, groups, stream=None, host=False):
        """Several joint batches side by side (tpamd_time_joint_groups_*): `groups` is a list of
        dicts with keys inputs, outputs, num_samples and optional safety, max_solver_loops -- each
        what time_joint_paths takes."""
        G = len(groups)
        bts, jis, pos = (_JointBatch * G)(), (_JointInputs * G)(), (_PathOutputs * G)()
        for g, grp in enumerate(groups):
            cp = grp["inputs"]["control_points"]
            B, P, D = cp.shape
            bts[g] = _JointBatch(B, D, int(grp["num_samples"]), P, int(grp.get("max_solver_loops", 0)),
                                 0, float(grp.get("safety", 0.8)))
            jis[g] = _JointInputs(*[_ptr(grp["inputs"].get(k)) for k in (
                "knots", "control_points", "max_velocity", "max_acceleration", "path_start", "delta",
                "sd_start", "sdd_start", "time_start", "num_samples_per_path")])
            pos[g] = _PathOutputs(*[_ptr(grp["outputs"].get(k)) for k in (
                "time", "s", "sd", "sdd", "q", "qd", "qdd", "last_extremal_index",
                "max_time_increment", "status", "sd2")])
        if host:
            _check(self._lib.tpamd_time_joint_groups_host(self._h, G, bts, jis, pos),
                   "tpamd_time_joint_groups_host")
        else:
            _check(self._lib.tpamd_time_joint_groups_device(self._h, G, bts, jis, pos,
                                                            _stream_ptr(stream)),
                   "tpamd_time_joint_groups_device")

    def sample_joint_paths(self, knots, control_points, path_start, delta, num_samples):
        """Host numpy arrays -> (q, q1, q2) [B][N][D]."""
        cp = np.ascontiguousarray(control_points, dtype=np.float64)
        kn = np.ascontiguousarray(knots, dtype=np.float64)
        B, P, D = cp.shape
        ps = np.ascontiguousarray(np.broadcast_to(path_start, (B,)), dtype=np.float64)
        dl = np.ascontiguousarray(np.broadcast_to(delta, (B,)), dtype=np.float64)
        out = [np.zeros((B, num_samples, D)) for _ in range(3)]
        _check(self._lib.tpamd_sample_joint_paths_host(
            self._h, B, D, int(num_samples), P, _ptr(kn), _ptr(cp), _ptr(ps), _ptr(dl),
            _ptr(out[0]), _ptr(out[1]), _ptr(out[2])), "tpamd_sample_joint_paths_host")
        return tuple(out)

    # ------------------------------------------------------- constraint rows
    def optimize_rows(self, inputs, outputs, max_solver_loops=0, stream=None, host=False):
        a = inputs["a"]
        B, N, Cn = a.shape
        bt = _RowsBatch(B, N, Cn, int(max_solver_loops))
        ri = _RowsInputs(*[_ptr(inputs.get(k)) for k in (
            "a", "b", "lower", "upper", "s_start", "s_end", "sd_start", "sdd_start",
            "time_start")])
        po = _PathOutputs(*[_ptr(outputs.get(k)) for k in (
            "time", "s", "sd", "sdd", "q", "qd", "qdd", "last_extremal_index",
            "max_time_increment", "status", "sd2")])
        if host:
            _check(self._lib.tpamd_optimize_rows_host(self._h, C.byref(bt), C.byref(ri),
                                                      C.byref(po)), "tpamd_optimize_rows_host")
        else:
            _check(self._lib.tpamd_optimize_rows_device(self._h, C.byref(bt), C.byref(ri),
                                                        C.byref(po), _stream_ptr(stream)),
                   "tpamd_optimize_rows_device")

    # ------------------------------------------------------- Cartesian paths
    def time_cartesian_paths(self, inputs, outputs, safety=0.8, max_solver_loops=0, stream=None,
                             host=False):
        """inputs: ik_positions [B][N][D], jacobians [B][N][6][D], max_velocity,
        max_acceleration [B][D], max_translational_velocity, max_rotational_velocity,
        path_start, delta, sd_start, time_start [B] (+ optional sdd_start). outputs as for
        time_joint_paths (q, if given, receives the IK positions)."""
        B, N, D = inputs["ik_positions"].shape
        bt = _CartesianBatch(B, D, N, int(max_solver_loops), float(safety))
        ci = _CartesianInputs(*[_ptr(inputs.get(k)) for k in _CARTESIAN_INPUT_KEYS])
        po = _PathOutputs(*[_ptr(outputs.get(k)) for k in (
            "time", "s", "sd", "sdd", "q", "qd", "qdd", "last_extremal_index",
            "max_time_increment", "status", "sd2")])
        if host:
            _check(self._lib.tpamd_time_cartesian_paths_host(self._h, C.byref(bt), C.byref(ci),
                                                             C.byref(po)),
                   "tpamd_time_cartesian_paths_host")
        else:
            _check(self._lib.tpamd_time_cartesian_paths_device(self._h, C.byref(bt), C.byref(ci),
                                                               C.byref(po), _stream_ptr(stream)),
                   "tpamd_time_cartesian_paths_device")

    def sample_pose_splines(self, knots, translation_points, rotation_points, path_start, delta,
                            num_samples):
        """Host numpy arrays: knots [B][P+3], translation [B][P][3], rotation [B][P][4] (w, x, y,
        z) -> poses [B][N][7] (tpamd_sample_pose_splines_host)."""
        kn = np.ascontiguousarray(knots, dtype=np.float64)
        tr = np.ascontiguousarray(translation_points, dtype=np.float64)
        ro = np.ascontiguousarray(rotation_points, dtype=np.float64)
        B, P, _ = tr.shape
        ps = np.ascontiguousarray(np.broadcast_to(path_start, (B,)), dtype=np.float64)
        dl = np.ascontiguousarray(np.broadcast_to(delta, (B,)), dtype=np.float64)
        out = np.zeros((B, int(num_samples), 7))
        _check(self._lib.tpamd_sample_pose_splines_host(self._h, B, int(num_samples), P, _ptr(kn),
                                                        _ptr(tr), _ptr(ro), _ptr(ps), _ptr(dl),
                                                        _ptr(out)), "tpamd_sample_pose_splines_host")
        return out

    def find_max_sd2(self, a, b, lower, upper):
        """Host numpy [num][C] -> (sd2max, sddmax, sd2zero) [num]."""
        a, b, lower, upper = (np.ascontiguousarray(x, dtype=np.float64)
                              for x in (a, b, lower, upper))
        num, Cn = a.shape
        o = [np.zeros(num) for _ in range(3)]
        _check(self._lib.tpamd_find_max_sd2_host(self._h, num, Cn, _ptr(a), _ptr(b), _ptr(lower),
                                                 _ptr(upper), _ptr(o[0]), _ptr(o[1]), _ptr(o[2])),
               "tpamd_find_max_sd2_host")
        return tuple(o)

    def query(self, time, s, sd, status, t_query, out_s, out_sd, out_sdd, ok=None, stream=None,
              sd2=None):
        """sd2: the solve's squared velocities (outputs["sd2"]); None = the engine's copy from
        its last solve (TpamdError "stale" if time is not that solve's output)."""
        B, N = time.shape
        K = t_query.shape[1]
        _check(self._lib.tpamd_query_device(self._h, B, N, K, _ptr(time), _ptr(s), _ptr(sd),
                                            _ptr(sd2), _ptr(status), _ptr(t_query), _ptr(out_s),
                                            _ptr(out_sd), _ptr(out_sdd), _ptr(ok),
                                            _stream_ptr(stream)), "tpamd_query_device")

    def resample_uniform(self, sol, max_acceleration, start_sec, time_step, out, stream=None,
                         skip=False):
        """sol: dict time,s,sd,sdd [B][N], q,qd,qdd [B][N][D], status. out: dict out_time..
        [B][max_out], out_q.. [B][max_out][D], count [B] int32. skip=True: the
        kSkipSamplesCloserThanTimeStep method instead of the uniform one."""
        B, N, D = sol["q"].shape
        max_out = out["out_time"].shape[1]
        args = _ResampleArgs(
            B, N, D, max_out,
            *[_ptr(sol[k]) for k in ("time", "s", "sd", "sdd", "q", "qd", "qdd")],
            _ptr(max_acceleration), _ptr(start_sec), float(time_step), _ptr(sol.get("status")),
            *[_ptr(out[k]) for k in ("out_time", "out_s", "out_sd", "out_sdd", "out_q",
                                     "out_qd", "out_qdd", "count")])
        fn = self._lib.tpamd_resample_skip_device if skip else self._lib.tpamd_resample_uniform_device
        _check(fn(self._h, C.byref(args), _stream_ptr(stream)),
               "tpamd_resample_skip_device" if skip else "tpamd_resample_uniform_device")

    def debug_boundary(self, B, N):
        arr = {k: np.zeros((B, N)) for k in ("sd2_max", "sdd_max", "sdd_min", "sd2_zero", "sd2")}
        arr["type"] = np.zeros((B, N), dtype=np.uint8)
        _check(self._lib.tpamd_debug_copy_boundary(
            self._h, B, N, _ptr(arr["sd2_max"]), _ptr(arr["sdd_max"]), _ptr(arr["sdd_min"]),
            _ptr(arr["sd2_zero"]), _ptr(arr["type"]), _ptr(arr["sd2"])),
            "tpamd_debug_copy_boundary")
        return arr

    def debug_keep_boundary(self, on=True):
        """Have the fused joint sweep store sdd_max/sdd_min/type for debug_boundary()."""
        self._lib.tpamd_debug_keep_boundary(self._h, 1 if on else 0)

    def rebuild_time(self, sd, ds, time_start, out, num_shards, paths_per_shard, num_samples,
                     shard_stride, num_samples_per_path=None, stream=None):
        """time [num_shards * paths_per_shard][N] from sd (tpamd_rebuild_time_device). sd, ds and
        time_start are tensors (views) that start at shard 0's data; shard r lies shard_stride
        doubles further."""
        _check(self._lib.tpamd_rebuild_time_device(
            self._h, num_shards, paths_per_shard, num_samples, shard_stride, _ptr(sd), _ptr(ds),
            _ptr(time_start), _ptr(num_samples_per_path), _ptr(out), _stream_ptr(stream)),
            "tpamd_rebuild_time_device")

    def debug_kernel_vgprs(self, which):
        """Registers per lane of the 7-joint sampling/LP kernel (0) / sweep kernel (1)."""
        n = self._lib.tpamd_debug_kernel_vgprs(self._h, which)
        _check(min(n, 0), "tpamd_debug_kernel_vgprs")
        return n

    def debug_diag(self, B):
        out = np.zeros((B, 64), dtype=np.int64)
        _check(self._lib.tpamd_debug_copy_diag(self._h, B, _ptr(out)), "tpamd_debug_copy_diag")
        return out

    # ---------------------------------------------------------------- timing
    def profile_enable(self, on=True):
        """True / 1: events around every kernel; 2: around the sweep kernel only; False: off."""
        self._lib.tpamd_profile_enable(self._h, int(on) if on in (1, 2) else (1 if on else 0))

    def profile_reset(self):
        self._lib.tpamd_profile_reset(self._h)

    def profile_mean_ms(self, kernel_index):
        n = C.c_int(0)
        ms = self._lib.tpamd_profile_mean_ms(self._h, int(kernel_index), C.byref(n))
        return ms, n.value

    def profile_summary(self):
        out = {}
        for k in range(self._lib.tpamd_profile_num_kernels()):
            ms, n = self.profile_mean_ms(k)
            out[self._lib.tpamd_profile_kernel_name(k).decode()] = (ms, n)
        return out


def _stream_ptr(stream):
    if stream is None:
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if isinstance(stream, int):
        return C.c_void_p(stream)
    return C.c_void_p(stream.cuda_stream)


def alloc_joint_outputs(B, N, D, device, with_q=True, with_derivs=True):
    import torch
    f = dict(dtype=torch.float64, device=device)
    out = dict(time=torch.empty(B, N, **f), s=torch.empty(B, N, **f), sd=torch.empty(B, N, **f),
               sdd=torch.empty(B, N, **f),
               last_extremal_index=torch.zeros(B, dtype=torch.int32, device=device),
               max_time_increment=torch.zeros(B, **f),
               status=torch.full((B,), -1, dtype=torch.int32, device=device))
    if with_q:
        out["q"] = torch.empty(B, N, D, **f)
    if with_derivs:
        out["qd"] = torch.empty(B, N, D, **f)
        out["qdd"] = torch.empty(B, N, D, **f)
    return out


def upload_joint_batch(batch, device):
    import torch
    keys = ("knots", "control_points", "vmax", "amax", "path_start", "delta", "sd_start",
            "time_start")
    t = {k: torch.from_numpy(np.ascontiguousarray(batch[k])).to(device) for k in keys}
    return dict(knots=t["knots"], control_points=t["control_points"], max_velocity=t["vmax"],
                max_acceleration=t["amax"], path_start=t["path_start"], delta=t["delta"],
                sd_start=t["sd_start"], time_start=t["time_start"])
