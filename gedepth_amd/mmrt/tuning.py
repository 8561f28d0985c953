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
    import tempfile
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
    # one scratch copy per committed content, user and rank: an updated db never hides behind a stale copy
    dst = os.path.join(tempfile.gettempdir(), f'gedepth_amd_miopen_db_{digest.hexdigest()[:10]}_{os.getuid()}_'
                                              f'{os.environ.get("LOCAL_RANK", "0")}')
    os.makedirs(dst, exist_ok=True)
    for f in files:
        if not os.path.isfile(os.path.join(dst, f)):
            shutil.copy(os.path.join(path, f), os.path.join(dst, f))
    os.environ['MIOPEN_USER_DB_PATH'] = dst
    return True
