"""MIOpen cold-start control.  The image ships NO gfx950 kernel database or find-db
(torch/share/miopen/db has gfx942/gfx90a only), so every new convolution shape is
JIT-compiled on first use: measured on a fresh MI355X box, the first BCNN step at
batch 64 / 448x448 costs 20 s (forward) + 253 s (backward) before a single image is
trained (419 s with channels_last tensors); with the find results cached the same step starts in ~2 s.  MIOpen keeps what it
compiled in $MIOPEN_CUSTOM_CACHE_DIR (kernel binaries, *.ukdb) and
$MIOPEN_USER_DB_PATH (find db).  hawkeye_amd/miopen_db/ holds a ~200 KB cache
populated by `tools/warm_miopen.sh` on a GPU box for the shapes bench.py / smoke()
use; it is copied to a scratch directory (MIOpen writes into it) and the two
variables are pointed there.  Pure plumbing: no effect on numerics or on the HIP kernels;
if the seed is missing or stale MIOpen simply compiles again.
"""
import os
import shutil
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = os.path.join(HERE, 'miopen_db')


def use_in_tree_cache(base=None):
    """Call before the first convolution runs.  Respects variables the user already set.
    HAWKEYE_MIOPEN_DIR=<dir> uses <dir> directly (how the seed is (re)populated)."""
    if 'MIOPEN_CUSTOM_CACHE_DIR' in os.environ and 'MIOPEN_USER_DB_PATH' in os.environ:
        return os.environ['MIOPEN_CUSTOM_CACHE_DIR']
    base = base or os.environ.get('HAWKEYE_MIOPEN_DIR')
    try:
        if base is None:
            # one scratch copy per rank: MIOpen's sqlite cache is not meant to be shared by concurrently starting processes
            base = os.path.join(tempfile.gettempdir(),
                                f"hawkeye_miopen_{os.getuid()}_r{os.environ.get('LOCAL_RANK', '0')}")
            for sub in ('cache', 'db'):
                dst, src = os.path.join(base, sub), os.path.join(SEED, sub)
                os.makedirs(dst, exist_ok=True)
                if os.path.isdir(src):
                    for f in os.listdir(src):
                        if not os.path.exists(os.path.join(dst, f)):
                            shutil.copy2(os.path.join(src, f), os.path.join(dst, f))
        else:
            os.makedirs(os.path.join(base, 'cache'), exist_ok=True)
            os.makedirs(os.path.join(base, 'db'), exist_ok=True)
    except OSError:
        return None
    # MIOpen's first-use "find" times every applicable solver, including the naive reference convolutions
    # (measured in the find db: 0.3-1.8 s PER RUN at the bench shapes vs 4-20 ms for the real solvers) - that, not
    # compilation, is what makes a cold start take minutes.  They are never selected; keep them out of the search.
    for k in ('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', 'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD',
              'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW'):
        os.environ.setdefault(k, '0')
    os.environ.setdefault('MIOPEN_CUSTOM_CACHE_DIR', os.path.join(base, 'cache'))
    os.environ.setdefault('MIOPEN_USER_DB_PATH', os.path.join(base, 'db'))
    _warn_if_seed_is_for_another_miopen()
    return base


def seed_miopen_version():
    """(major, minor, patch) the in-tree find-db was written by, from its file name (`gfx950100.HIP.3_5_0_<build>.ufdb.txt`:
    MIOpen names the user find-db after its own version and only ever opens the file of the running version)."""
    import re
    try:
        for f in os.listdir(os.path.join(SEED, 'db')):
            m = re.search(r'\.HIP\.(\d+)_(\d+)_(\d+)_', f)
            if m:
                return tuple(int(g) for g in m.groups())
    except OSError:
        pass
    return None


def _warn_if_seed_is_for_another_miopen():
    """The seed goes stale silently on another ROCm: MIOpen looks for a find-db named after ITS version, does not find one,
    and compiles / searches from scratch (minutes on the first step).  Say so once instead."""
    try:
        import torch
        have = torch.backends.cudnn.version() if torch.cuda.is_available() else None     # MIOpen: major * 10^6 + minor * 10^3 + patch
    except Exception:  # noqa: BLE001
        have = None
    seed = seed_miopen_version()
    if have is None or seed is None:
        return
    running = (have // 1000000, (have // 1000) % 1000, have % 1000)
    if running[:2] != seed[:2]:
        import warnings
        warnings.warn(f'hawkeye_amd/miopen_db was populated by MIOpen {seed[0]}.{seed[1]}.{seed[2]}, this process runs MIOpen '
                      f'{running[0]}.{running[1]}.{running[2]}: the cached find results will not be used - the first training step '
                      f'searches and compiles every convolution again (re-populate with tools/warm_miopen.sh)', stacklevel=2)
