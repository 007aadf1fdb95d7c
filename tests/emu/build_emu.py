"""TEST HARNESS: compiles the UNMODIFIED engine sources (spotlight_amd/csrc/*.hip) with g++
against the fiber emulator (tests/emu/hip/hip_runtime.h) into tests/emu/_build/libspotlight_emu.so.
Never used by spotlight_amd itself."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'spotlight_amd', 'csrc')
# SLK_EMU_CXXFLAGS: extra -D flags (the kernels' build-time variants, e.g. -DSLK_ITEM_KEYPF=1), each set built into its own directory
EXTRA = os.environ.get('SLK_EMU_CXXFLAGS', '').split()
OUT = os.path.join(HERE, '_build' + (''.join(c if c.isalnum() else '_' for c in '_'.join(EXTRA)) if EXTRA else ''))
LIB = os.path.join(OUT, 'libspotlight_emu.so')


def build(force=False):
    """Serialised across processes (pytest-xdist workers would otherwise compile into the same files at once)."""
    import fcntl
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build(force)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build(force):
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))
    deps = srcs + [os.path.join(CSRC, 'slk_common.h'), os.path.join(CSRC, 'slk_kernels.h'), os.path.join(ROOT, 'include', 'spotlight_hip.h'),
                   os.path.join(HERE, 'emu_runtime.cpp'), os.path.join(HERE, 'hip', 'hip_runtime.h')]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    objs, procs = [], []
    for s in srcs + [os.path.join(HERE, 'emu_runtime.cpp')]:
        o = os.path.join(OUT, os.path.basename(s) + '.o')
        objs.append(o)
        cmd = ['g++', '-std=c++17', '-O1', '-g', '-fPIC', '-fvisibility=hidden', '-ffp-contract=off',
               '-Wall', '-Wno-unused-function', '-Wno-unknown-pragmas'] + EXTRA + ['-I', HERE, '-x', 'c++', '-c', s, '-o', o]
        procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError('g++ failed on %s' % s)
    subprocess.check_call(['g++', '-shared', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build(force=True))
