"""Build libbm355.so (HIP, gfx950) in-tree with hipcc.

The shared library is the product: there is no CPU fallback.  `build()` is
called by `__graft_entry__.build()` and lazily by `_ffi.load()` when the
library is missing or older than its sources.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libbm355.so')
SOURCES = ['bm355.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
         '-ffp-contract=off',          # every fma is an explicit fmaf (DESIGN.md "Numerics")
         '-mllvm', '-amdgpu-mfma-vgpr-form',   # MFMA accumulators stay in VGPRs: without it hipcc parks the FP64
                                               # accumulators in AGPRs and copies all 16 of them in and out around
                                               # every K chunk (32 VALU instructions per 16 MFMAs)
         '-Wall', '-Wno-unused-function']


def _sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _deps():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(HERE, '..', 'include', 'bm355.h'))
    return deps


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps() if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: cannot build libbm355.so')
    # one process per GPU may reach this point at the same time (torch.distributed.run): build
    # under an exclusive lock into a temporary file and rename, re-checking once the lock is held
    import fcntl
    with open(os.path.join(HERE, '.build.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build():
            return LIB
        tmp = LIB + '.tmp.%d' % os.getpid()
        cmd = [hipcc] + FLAGS + _sources() + ['-o', tmp]
        if verbose:
            print(' '.join(cmd))
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError('hipcc failed:\n' + r.stdout)
        os.replace(tmp, LIB)
        if verbose and r.stdout:
            print(r.stdout)
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
