"""Library-kernel selection caches for the parts of the step that stay in rocBLAS / hipBLASLt / MIOpen.

The Linear layers of the Swin encoder and of the HAHI attention run as library GEMMs (plain GEMMs are not worth a
hand-written kernel).  hipBLASLt's default heuristic picks poor macro-tiles for the tall-skinny shapes of this
model (M = 8 x 24640 ... 8 x 98560 tokens, K and N of 96 ... 768): PyTorch's TunableOp times the candidate solutions
once and records the winner per shape.  The winners for the BASELINE workloads on gfx950 are committed in
``gedepth_amd/tuning/tunableop_gfx950.csv`` and only *looked up* at run time (no tuning, no start-up cost); shapes that
are not in the file use the library default.  ``tools/tune_tables.py`` regenerates both tables.
"""
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
GEMM_TABLE = os.environ.get('GE_GEMM_TABLE') or os.path.join(os.path.dirname(_HERE), 'tuning', 'tunableop_gfx950.csv')
MIOPEN_DB = os.path.join(os.path.dirname(_HERE), 'tuning', 'miopen')


def use_tuned_gemms(mode='load', path=GEMM_TABLE):
    """mode: 'off' | 'load' (look up the committed table) | 'tune' (time unseen shapes and append them on exit)."""
    if mode not in ('off', 'load', 'tune'):
        raise ValueError(f'gemm tuning mode {mode!r}')
    import torch
    import torch.cuda.tunable as tunable
    if not torch.cuda.is_available():                  # TunableOp state lives in the HIP context
        return False
    if mode == 'off':
        tunable.enable(False)
        return False
    tunable.enable(True)
    tunable.tuning_enable(mode == 'tune')
    tunable.set_filename(path, insert_device_ordinal=False)
    if os.path.isfile(path):
        tunable.read_file(path)
    elif mode == 'load':
        tunable.enable(False)
        return False
    return True


def use_miopen_find_db(path=MIOPEN_DB):
    """Seed MIOpen's user find-db with the committed one (call before the first convolution, i.e. before MIOpen creates
    its handle).  With the db in place ``torch.backends.cudnn.benchmark = True`` resolves every convolution of the BASELINE
    workloads from the db (+6 % step rate) instead of timing all solvers for ~5 minutes on a fresh machine.  The db is
    copied to a scratch directory because MIOpen appends to it while running."""
    import shutil
    skip_naive_conv_solvers()
    if os.environ.get('MIOPEN_USER_DB_PATH'):
        return True                                    # the user manages the db
    files = [f for f in (os.listdir(path) if os.path.isdir(path) else []) if f.endswith('db.txt')]
    if not files:
        return False
    import hashlib
    digest = hashlib.sha1()
    for f in sorted(files):
        with open(os.path.join(path, f), 'rb') as fh:
            digest.update(fh.read())
    # one private copy per committed content and rank under a user-owned 0700 cache directory (never the shared tempdir:
    # a predictable path there could be pre-created by another local user and MIOpen reads AND appends to it); symlinks are
    # refused and the copied files are always rewritten
    cache = os.environ.get('XDG_CACHE_HOME') or os.path.join(os.path.expanduser('~'), '.cache')
    base = os.path.join(cache, 'gedepth_amd')
    dst = os.path.join(base, f'miopen_db_{digest.hexdigest()[:10]}_{os.environ.get("LOCAL_RANK", "0")}')
    for d in (base, dst):
        os.makedirs(d, mode=0o700, exist_ok=True)
        st = os.lstat(d)
        import stat
        if stat.S_ISLNK(st.st_mode) or st.st_uid != os.getuid():
            raise RuntimeError(f'{d} is a symlink or owned by another user; refusing to use it as the MIOpen user db')
        os.chmod(d, 0o700)
    for f in files:
        target = os.path.join(dst, f)
        if os.path.islink(target):
            os.unlink(target)
        tmp = target + f'.{os.getpid()}.tmp'
        shutil.copyfile(os.path.join(path, f), tmp)
        os.replace(tmp, target)
    os.environ['MIOPEN_USER_DB_PATH'] = dst
    return True


def skip_naive_conv_solvers():
    """Keep MIOpen's reference ('naive') direct convolutions out of the find step.  On a machine without a kernel cache MIOpen
    re-times every applicable solver per problem whatever the find-db says; the naive NHWC kernels take 0.1 - 7 s per call (150 s of a
    160 s start-up on MI355X, profiles/README.md) and never win.  They stay available if the user sets the variables."""
    for d in ('FWD', 'BWD', 'WRW'):
        os.environ.setdefault(f'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_{d}', '0')
