"""Build tests/emu/_build/libhawkeye_emu.so: the kernel sources of hawkeye_amd/csrc compiled for x86 against the
CPU emulation of the HIP device model in tests/emu/shim (see the header there).  TEST INFRASTRUCTURE ONLY - the
product library is hawkeye_amd/csrc/libhawkeye_hip.so (gfx950) and nothing under hawkeye_amd/ loads this one.
    python tests/emu/build_emu.py            # incremental; prints the path
"""
import glob
from concurrent.futures import ThreadPoolExecutor
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'hawkeye_amd', 'csrc')
# HK_EMU_ASAN=1: AddressSanitizer build (out-of-bounds accesses of inputs / outputs / workspaces / LDS abort with a
# report).  Python must then be started with LD_PRELOAD=asan_runtime() and ASAN_OPTIONS=detect_leaks=0 - see
# tests/emu/README.md.
ASAN = os.environ.get('HK_EMU_ASAN') == '1'
OUT = os.path.join(HERE, '_build_asan' if ASAN else '_build')
LIB = os.path.join(OUT, 'libhawkeye_emu.so')


def asan_runtime():
    import glob as _g
    hits = _g.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so')
    return hits[0] if hits else None


def _compiler():
    for c in ('/opt/rocm/lib/llvm/bin/clang++', shutil.which('clang++') or ''):   # ext_vector_type needs clang
        if c and os.path.exists(c):
            return c
    return None


def build(verbose=False):
    cxx = _compiler()
    if cxx is None:
        raise RuntimeError('no clang++ found: the kernels use ext_vector_type')
    os.makedirs(OUT, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    deps = srcs + glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(HERE, 'shim', 'hip', '*.h')) + glob.glob(os.path.join(HERE, 'shim', '*.S')) + \
        glob.glob(os.path.join(ROOT, 'include', '*.h'))
    if os.path.exists(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(d) for d in deps):
        return LIB
    san = ['-fsanitize=address', '-shared-libasan', '-fno-omit-frame-pointer', '-g'] if ASAN else []
    flags = san + ['-x', 'c++', '-std=c++17', '-O1', '-fPIC', '-ffp-contract=off', '-Wno-unknown-attributes', '-Wno-unused-value',
             '-Wno-pass-failed', '-Wno-unknown-pragmas', '-Wno-psabi',
             '-I' + os.path.join(HERE, 'shim'), '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC]
    def compile_one(s):
        o = os.path.join(OUT, os.path.basename(s) + '.o')
        cmd = [cxx] + flags + ['-c', s, '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return o

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, srcs))
    sw = os.path.join(OUT, 'hipemu_switch.o')
    subprocess.run([cxx, '-c', os.path.join(HERE, 'shim', 'hipemu_switch.S'), '-o', sw], check=True)
    subprocess.run([cxx, '-shared', '-Wl,-Bsymbolic'] + (['-fsanitize=address', '-shared-libasan'] if ASAN else []) +
                   ['-o', LIB] + objs + [sw], check=True)   # never bind to the gfx950 library's symbols
    return LIB


if __name__ == '__main__':
    print(build(verbose='-v' in sys.argv))
