"""ctypes binding of libnagp.so (include/nagp.h).  No torch types cross this boundary.

The library is built in-tree by `__graft_entry__.build()` (hipcc --offload-arch=gfx950).  There is
no CPU fallback: if the shared object is missing or no GPU is visible every entry point raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(PKG_ROOT, 'libnagp.so')
CSRC = os.path.join(PKG_ROOT, 'csrc')
INCLUDE = os.path.join(os.path.dirname(PKG_ROOT), 'include')
EXTRA_LINK = ['-L/opt/rocm/lib', '-lrccl', '-lpthread']   # nagp_batch_run: ncclAllReduce of nlZ over the GPUs of the node

NAGP_OK = 0
KIND_GF_EP, KIND_IHGP, KIND_GIEKF = 0, 1, 2
MODE_PREDICT, MODE_NLML = 0, 1
LIK_POWER, LIK_POWER_NMF, LIK_POWER_NMF_SQRT = 0, 1, 2
LINK_SOFTPLUS, LINK_EXP = 0, 1
FLAG_IHGP_CONSTRAINTS, FLAG_EKF_RESET_P, FLAG_WANT_PS, FLAG_MIXTURE_RULE = 0x1, 0x2, 0x4, 0x8
N_KERNELS = 8
KERNEL_NAMES = ['filter', 'gain', 'scan', 'epsite', 'reduce', 'filter_lin', 'output', 'other']

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_lp = C.POINTER(C.c_int64)


class Model(C.Structure):
    _fields_ = [('S', C.c_int32), ('M', C.c_int32), ('D', C.c_int32), ('N', C.c_int32),
                ('block_offsets', c_ip), ('A', c_dp), ('Q', c_dp), ('Pinf', c_dp), ('h_val', c_dp),
                ('Wnmf', c_dp), ('lik_param', C.c_double)]


class IhgpTables(C.Structure):
    _fields_ = [('n_grid', C.c_int32), ('r_grid', c_dp), ('PPlist', c_dp), ('pp_offsets', c_lp),
                ('PGlist', c_dp), ('pg_offsets', c_lp)]


class Opts(C.Structure):
    _fields_ = [('kind', C.c_int32), ('mode', C.c_int32), ('lik_kind', C.c_int32), ('link_kind', C.c_int32),
                ('link_shift', C.c_double), ('n_pts', C.c_int32), ('cub_dim', C.c_int32), ('wn', c_dp),
                ('xn_unscaled', c_dp), ('ep_fraction', C.c_double), ('ep_itts', C.c_int32),
                ('ep_damping', c_dp), ('l_iter', C.c_int32), ('predict_at_k1', C.c_int32),
                ('flags', C.c_uint32), ('device', C.c_int32), ('chunk', C.c_int32), ('ttau0', c_dp), ('tnu0', c_dp)]


class Out(C.Structure):
    _fields_ = [('Eft', c_dp), ('Varft', c_dp), ('MS', c_dp), ('PS', c_dp), ('ttau', c_dp), ('tnu', c_dp),
                ('R', c_dp), ('lZ', c_dp), ('nlZ', c_dp), ('maxDiffM', c_dp), ('maxDiffP', c_dp),
                ('counters', c_lp), ('MF', c_dp)]


class Timings(C.Structure):
    _fields_ = [('ms', C.c_double * N_KERNELS), ('launches', C.c_int64 * N_KERNELS), ('total_ms', C.c_double)]


EXPORTS = ['nagp_version', 'nagp_device_count', 'nagp_strerror', 'nagp_last_error', 'nagp_ep_run',
           'nagp_ihgp_run', 'nagp_giekf_run', 'nagp_plan_create', 'nagp_plan_upload_y', 'nagp_plan_execute',
           'nagp_plan_timings', 'nagp_plan_download', 'nagp_plan_device_bytes', 'nagp_plan_destroy', 'nagp_plan_upload_sites',
           'nagp_batch_partition', 'nagp_batch_run', 'nagp_shutdown', 'nagp_reconstruct', 'nagp_mom_eval', 'nagp_iekf_update1', 'nagp_fastfb_run',
           'nagp_giekf_nlml_grad']


class NagpError(RuntimeError):
    pass


def source_hash():
    """SHA-256 over the sources libnagp.so is built from (csrc/*, include/nagp.h), in name order."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(f.encode()); h.update(fh.read())
    with open(os.path.join(INCLUDE, 'nagp.h'), 'rb') as fh:
        h.update(b'nagp.h'); h.update(fh.read())
    return h.hexdigest()


HASH_PATH = LIB_PATH + '.srchash'


LIB_ASAN_PATH = os.path.join(PKG_ROOT, 'libnagp_asan.so')


# translation units that hold HOST code of the C ABI (argument validation, packing): instrumented in the ASan build
ASAN_UNITS = ('nagp_api.hip', 'nagp_grad.hip')


def _tu_hash(f, extra=''):
    """hash of one translation unit: its own text, every header of csrc/ and include/nagp.h, the flags"""
    import hashlib
    h = hashlib.sha256(extra.encode())
    # (nagp_api_*.hpp are the parts of the ONE host translation unit nagp_api.hip: no other unit includes them)
    names = [f] + sorted(x for x in os.listdir(CSRC) if x.endswith(('.hpp', '.h')) and (f == 'nagp_api.hip' or not x.startswith('nagp_api_')))
    for n in names:
        with open(os.path.join(CSRC, n), 'rb') as fh:
            h.update(n.encode()); h.update(fh.read())
    with open(os.path.join(INCLUDE, 'nagp.h'), 'rb') as fh:
        h.update(fh.read())
    return h.hexdigest()[:16]


def build(force=False, verbose=False, jobs=None, asan=False):
    """Compile csrc/*.hip -> libnagp.so for gfx950 (cross-compiles without a GPU): one object per translation unit
    (nagp_api.hip = host code, inst_*.hip = groups of kernel instantiations), compiled in parallel into a per-unit object
    cache (build/obj), then linked.  The library is stale when the hash of its sources differs from the one recorded
    next to it at build time (mtimes are not trusted).
    asan=True: libnagp_asan.so with the HOST code of the C ABI (nagp_api.hip) under AddressSanitizer -- the kernels are
    the ordinary objects (no GPU sanitizer on this pool); used by the CPU tests of the argument-validation paths."""
    want = source_hash()
    lib_path = LIB_ASAN_PATH if asan else LIB_PATH
    hash_path = lib_path + '.srchash'
    if not force and os.path.exists(lib_path) and os.path.exists(hash_path):
        with open(hash_path) as fh:
            if fh.read().strip() == want:
                return lib_path
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objdir = os.path.join(PKG_ROOT, 'build', 'obj')
    os.makedirs(objdir, exist_ok=True)
    tus = sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))
    flags = ['-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-I', INCLUDE]
    san = ['-fsanitize=address', '-fno-omit-frame-pointer', '-g', '-shared-libsan'] if asan else []

    def compile_one(f):
        fl = flags + (san if f in ASAN_UNITS else [])
        obj = os.path.join(objdir, '%s-%s.o' % (f[:-4], _tu_hash(f, ' '.join(fl))))
        if os.path.exists(obj) and not force:
            return f, obj, None, None
        tmp = obj + '.tmp.%d' % os.getpid()
        cmd = [hipcc] + fl + ['-c', '-o', tmp, os.path.join(CSRC, f)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode == 0:
            os.replace(tmp, obj)
        return f, obj, cmd, r

    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    with ThreadPoolExecutor(max_workers=jobs or max(1, min(usable, 8))) as ex:
        res = list(ex.map(compile_one, tus))
    bad = [x for x in res if x[3] is not None and x[3].returncode != 0]
    if verbose or bad:
        for f, obj, cmd, r in (bad or [x for x in res if x[3] is not None]):
            print(' '.join(cmd)); print(r.stdout); print(r.stderr)
    if bad:
        raise NagpError('hipcc failed on %s' % ', '.join(x[0] for x in bad))
    tmp = lib_path + '.tmp.%d' % os.getpid()
    cmd = [hipcc, '--offload-arch=gfx950', '-fPIC', '-shared', '-o', tmp] + [x[1] for x in res] + EXTRA_LINK + (['-fsanitize=address', '-shared-libsan'] if asan else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(' '.join(cmd)); print(r.stdout); print(r.stderr)
    if r.returncode != 0:
        raise NagpError('hipcc failed linking %s' % os.path.basename(lib_path))
    os.replace(tmp, lib_path)
    with open(hash_path, 'w') as fh:
        fh.write(want + '\n')
    keep = {os.path.basename(x[1]) for x in res}                    # drop objects of older sources (both variants keep theirs)
    for o in os.listdir(objdir):
        stem = o.rsplit('-', 1)[0]
        if o.endswith('.o') and o not in keep and not (stem + '.hip' in ASAN_UNITS):
            os.remove(os.path.join(objdir, o))
    return lib_path


_lib = None


def lib():
    """Load libnagp.so (raises NagpError when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get('NAGP_LIB') or LIB_PATH        # NAGP_LIB: developer override (same-box A/B of several builds, tools/ab_libs.sh)
    if not os.path.exists(path):
        raise NagpError('libnagp.so is missing (%s): run __graft_entry__.build() -- there is no CPU fallback' % path)
    L = C.CDLL(path)
    L.nagp_version.restype = C.c_int
    L.nagp_device_count.restype = C.c_int
    L.nagp_strerror.restype = C.c_char_p; L.nagp_strerror.argtypes = [C.c_int]
    L.nagp_last_error.restype = C.c_char_p
    L.nagp_ep_run.argtypes = [C.POINTER(Model), c_dp, C.c_int64, C.POINTER(Opts), C.POINTER(Out)]
    L.nagp_ihgp_run.argtypes = [C.POINTER(Model), C.POINTER(IhgpTables), c_dp, C.c_int64, C.POINTER(Opts), C.POINTER(Out)]
    L.nagp_giekf_run.argtypes = [C.POINTER(Model), c_dp, C.c_int64, C.POINTER(Opts), C.POINTER(Out)]
    L.nagp_plan_create.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(Model), C.POINTER(IhgpTables),
                                   C.c_int64, C.POINTER(Opts)]
    L.nagp_mom_eval.argtypes = [C.POINTER(Opts), C.c_int32, C.c_int32, c_dp, C.c_double, C.c_int64, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp]
    L.nagp_iekf_update1.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), c_dp, c_dp, C.c_double, C.c_double, C.c_int32,
                                    c_dp, c_dp, c_dp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32]
    L.nagp_fastfb_run.argtypes = [C.c_int32, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, C.c_int64, c_dp, C.POINTER(C.c_double), C.c_int32]
    L.nagp_giekf_nlml_grad.argtypes = [C.c_int32, C.POINTER(Model), C.POINTER(c_dp), C.c_int64, C.c_int32, C.POINTER(c_dp), C.POINTER(c_dp), C.POINTER(c_dp),
                                       c_dp, c_ip, c_ip, c_ip, c_dp, c_dp, C.c_int32]
    L.nagp_giekf_nlml_grad.restype = C.c_int
    L.nagp_plan_upload_y.argtypes = [C.c_void_p, C.POINTER(c_dp)]
    L.nagp_plan_upload_sites.argtypes = [C.c_void_p, C.POINTER(c_dp), C.POINTER(c_dp)]
    L.nagp_batch_partition.argtypes = [C.c_int32, C.c_int32, c_ip]
    L.nagp_batch_run.argtypes = [C.c_int32, C.POINTER(Model), C.POINTER(IhgpTables), C.POINTER(c_dp), C.c_int64, C.POINTER(Opts), C.POINTER(Out), C.c_int32, c_dp]
    L.nagp_shutdown.restype = None
    L.nagp_reconstruct.argtypes = [C.c_int32, C.c_int32, C.c_int64, c_dp, c_dp, c_dp, C.c_int32, C.c_double, C.c_int32, c_dp, c_dp,
                                   C.c_int32, C.c_uint64, c_dp, c_dp, c_dp, c_dp, C.c_int32]
    L.nagp_reconstruct.restype = C.c_int
    L.nagp_plan_execute.argtypes = [C.c_void_p]
    L.nagp_plan_timings.argtypes = [C.c_void_p, C.POINTER(Timings)]
    L.nagp_plan_download.argtypes = [C.c_void_p, C.POINTER(Out)]
    L.nagp_plan_device_bytes.argtypes = [C.c_void_p]; L.nagp_plan_device_bytes.restype = C.c_int64
    L.nagp_plan_destroy.argtypes = [C.c_void_p]; L.nagp_plan_destroy.restype = None
    for f in ('nagp_plan_upload_sites', 'nagp_batch_partition', 'nagp_batch_run', 'nagp_ep_run', 'nagp_ihgp_run', 'nagp_giekf_run', 'nagp_mom_eval', 'nagp_iekf_update1', 'nagp_fastfb_run', 'nagp_plan_create', 'nagp_plan_upload_y',
              'nagp_plan_execute', 'nagp_plan_timings', 'nagp_plan_download'):
        getattr(L, f).restype = C.c_int
    _lib = L
    return L


def check(status):
    if status != NAGP_OK:
        L = lib()
        raise NagpError('libnagp: %s (%d): %s' % (L.nagp_strerror(status).decode(), status, L.nagp_last_error().decode()))


def dptr(a):
    return a.ctypes.data_as(c_dp) if a is not None else c_dp()


def f64(a, order='F'):
    return np.require(np.asarray(a, dtype=np.float64), requirements=['A', 'O', 'W'] + (['F'] if order == 'F' else ['C']))
