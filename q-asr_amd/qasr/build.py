"""Builds the native library in-tree: hipcc --offload-arch=gfx950 -> qasr/libqasr_hip.so
(cross-compiles without a GPU; the .so travels to the GPU box with the snapshot)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), 'csrc')
# QASR_BUILD_TAG=<tag> (experiments): objects under build/obj_<tag>, library qasr/libqasr_<tag>.so - load it with QASR_LIB
TAG = os.environ.get('QASR_BUILD_TAG', '')
LIB = os.path.join(HERE, f'libqasr_{TAG}.so' if TAG else 'libqasr_hip.so')
SOURCES = ['qasr_kernels.hip', 'qasr_sep.hip', 'qasr_sep_t32.hip', 'qasr_sep_t32_dbg.hip', 'qasr_sep_t64.hip',
           'qasr_sep_t64_dbg.hip', 'qasr_sep_t128.hip', 'qasr_sep2_t32.hip', 'qasr_sep2_t32_dbg.hip', 'qasr_sep2_t64.hip', 'qasr_sep2_t64_dbg.hip', 'qasr_sep2_t128.hip', 'qasr_sep2_t128_dbg.hip', 'qasr_dense2.hip',
           'qasr_engine.hip', 'qasr_blob_check.cpp', 'qasr_frontend.hip', 'qasr_calib.hip', 'qasr_dynamic.hip', 'qasr_decoder.hip', 'qasr_stem.hip']


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + \
           [os.path.join(os.path.dirname(os.path.dirname(HERE)), 'include', 'qasr.h')]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + _headers()
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force=False, verbose=False):
    """Compiles what changed: an object is rebuilt when its source or any header is newer (or flags differ: QASR_HIPCC_FLAGS
    builds are always full), then everything is linked again."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    srcs = [os.path.join(CSRC, f) for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]
    extra = os.environ.get('QASR_HIPCC_FLAGS', '').split()          # experiments: -DSEP_WK=8 ...
    # one object per translation unit, compiled in parallel (the k_sep instantiations dominate), then one link
    objdir = os.path.join(os.path.dirname(HERE), 'build', 'obj_' + TAG if TAG else 'obj')
    os.makedirs(objdir, exist_ok=True)
    objs = [os.path.join(objdir, os.path.splitext(os.path.basename(f))[0] + '.o') for f in srcs]
    stamp = os.path.join(objdir, '.flags')
    flags_now = ' '.join(extra)
    if force or not os.path.exists(stamp) or open(stamp).read() != flags_now:
        todo = list(zip(srcs, objs))
    else:
        hdr_t = max(os.path.getmtime(h) for h in _headers())
        todo = [(s, o) for s, o in zip(srcs, objs)
                if not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_t)]

    def compile_one(src, obj):
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', '-o', obj] + extra + [src]
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)

    from concurrent.futures import ThreadPoolExecutor
    jobs = max(1, min(len(todo) or 1, (os.cpu_count() or 2)))
    with ThreadPoolExecutor(jobs) as ex:
        list(ex.map(lambda so: compile_one(*so), todo))
    with open(stamp, 'w') as f:
        f.write(flags_now)
    subprocess.run([hipcc, '--offload-arch=gfx950', '-fPIC', '-shared', '-o', LIB] + objs, check=True)
    return LIB


if __name__ == '__main__':
    print(build_native(force='--force' in sys.argv, verbose=True))
