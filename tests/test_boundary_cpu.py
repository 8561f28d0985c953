"""Drop-in boundary, host side (no GPU): config loader, registry/build protocol, state-dict schema,
C-ABI library exports.  SURVEY.md §8(b)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from gedepth_amd import hip
from gedepth_amd.depth.models import MODELS, build_depther
from gedepth_amd.mmrt.config import Config, DictAction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
REGISTRY_NAMES = ['DepthEncoderDecoder', 'DepthFormerSwin', 'HAHIHeteroNeck', 'LightPEMASKNeck', 'DynamicPENeckSOFT',
                  'DenseDepthHead', 'SigLoss', 'CrossEntropyLoss', 'BinaryCrossEntropyLoss']


def cfg_path(name):
    return os.path.join(ROOT, 'configs', 'depthformer', name)


def test_registry_names():
    for n in REGISTRY_NAMES:
        assert MODELS.get(n) is not None, n
    from gedepth_amd.mmrt.bricks import POSITIONAL_ENCODING
    assert POSITIONAL_ENCODING.get('SinePositionalEncoding') is not None


def test_config_inheritance_and_delete():
    cfg = Config.fromfile(cfg_path('depthformer_a.py'))
    assert cfg.model.backbone.embed_dims == 192 and cfg.model['backbone']['depths'] == [2, 2, 18, 2]
    assert cfg.model.dynamic_pe_neck.type == 'DynamicPENeckSOFT'
    assert cfg.model.backbone.drop_path_rate == 0.3             # inherited from the base
    assert [h['type'] for h in cfg.log_config.hooks] == ['TextLoggerHook', 'TensorboardLoggerHook']  # _delete_
    assert cfg.log_config.interval == 10
    assert cfg.lr_config.warmup_iters == 25600 and cfg.runner.max_iters == 76800
    cfg.merge_from_dict(DictAction.parse(['model.backbone.drop_path_rate=0.1', 'data.samples_per_gpu=4']))
    assert cfg.model.backbone.drop_path_rate == 0.1 and cfg.data.samples_per_gpu == 4
    assert 'model = dict(' in cfg.pretty_text


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree only exists in the build container')
@pytest.mark.parametrize('name', ['depthformer_v.py', 'depthformer_a.py', 'depthformer_a_ddad.py', 'depthformer_v_ddad.py'])
def test_reference_configs_load_unchanged_and_resolve_identically(name):
    """The reference's own config files load byte-unchanged through our loader, and our re-structured
    configs resolve to the same model / optimizer / schedule."""
    ref = Config.fromfile(os.path.join(REF, 'configs', 'depthformer', name))
    ours = Config.fromfile(cfg_path(name))
    for key in ['model', 'optimizer', 'optimizer_config', 'lr_config', 'runner', 'checkpoint_config', 'evaluation',
                'log_config', 'dist_params', 'workflow', 'data']:
        assert ref.to_dict()[key] == ours.to_dict()[key], key


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree only exists in the build container')
def test_reference_ddad_configs_load():
    for name in ['depthformer_v_ddad.py', 'depthformer_a_ddad.py']:
        cfg = Config.fromfile(os.path.join(REF, 'configs', 'depthformer', name))
        assert cfg.model.type == 'DepthEncoderDecoder'


def _build(name, **over):
    cfg = Config.fromfile(cfg_path(name))
    cfg.model.pretrained = None
    for k, v in over.items():
        cfg.model[k] = v
    return build_depther(cfg.model, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))


@pytest.mark.parametrize('cfg_name,fixture', [('depthformer_swint_v.py', 'e2e_T_V'), ('depthformer_swint_a.py', 'e2e_T_A'),
                                              ('depthformer_a.py', 'e2e_L_A')])
def test_state_dict_schema_matches_reference(cfg_name, fixture, golden):
    """Key names, order-insensitive, shapes and dtypes equal the state dict captured from the reference
    (516 entries for Swin-L-A)."""
    model = _build(cfg_name)
    spec = {k: tuple(s) for k, s in json.loads(str(golden(fixture)['spec']))}
    ours = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert set(ours) == set(spec), (sorted(set(ours) ^ set(spec))[:10])
    assert ours == spec
    if fixture == 'e2e_L_A':
        assert len(ours) == 516
    n_params = sum(p.numel() for p in model.parameters())
    assert n_params == sum(int(np.prod(spec[k])) for k, _ in model.named_parameters())


def test_build_depther_contract():
    cfg = Config.fromfile(cfg_path('depthformer_swint_v.py'))
    cfg.model.pretrained = None
    with pytest.raises(AssertionError):
        build_depther(cfg.model, train_cfg=dict())          # given twice
    with pytest.raises(KeyError):
        build_depther(dict(type='NoSuchDepther'))
    m = build_depther(cfg.model)
    assert m.test_cfg.mode == 'whole' and m.align_corners is True
    assert not m.dynamic_pe_neck_FLAGS and m.pe_mask_neck_FLAGS
    with pytest.raises(TypeError):
        m.forward_test(torch.zeros(1), [[]])                 # imgs must be a list (depther/base.py:74-77)
    with pytest.raises(ValueError):
        m.forward_test([torch.zeros(1)], [[], []])


def test_pretrained_pushed_into_backbone():
    cfg = Config.fromfile(cfg_path('depthformer_v.py'))
    assert cfg.model.pretrained.endswith('.pth')
    cfg.model.backbone.depths = [1, 1, 1, 1]
    m = build_depther(cfg.model)
    assert m.backbone.pretrained == cfg.model.pretrained


def test_init_weights_runs_and_is_deterministic_under_seed():
    torch.manual_seed(0)
    m = _build('depthformer_swint_a.py')
    m.init_weights()
    sd = m.state_dict()
    assert torch.count_nonzero(sd['neck.multi_att.sampling_offsets.weight']) == 0
    assert torch.count_nonzero(sd['neck.multi_att.attention_weights.bias']) == 0
    b = sd['neck.self_attn.sampling_offsets.bias'].view(8, 4, 8, 2)
    assert torch.allclose(b[0, 0, :, 0], torch.arange(1, 9.0)) and torch.allclose(b[0, 0, :, 1], torch.zeros(8), atol=1e-6)
    t = sd['backbone.stages.0.blocks.0.attn.w_msa.relative_position_bias_table']
    assert 0.005 < t.std() < 0.04


def test_cpu_tensors_fail_loudly():
    """There is no CPU/eager fallback in the product path."""
    from gedepth_amd.kernels import bilinear_resize, window_attention
    if not hip.is_built():
        pytest.skip('library not built')
    with pytest.raises(RuntimeError, match='MI355X only'):
        bilinear_resize(torch.zeros(1, 1, 4, 4), (8, 8))
    with pytest.raises(RuntimeError, match='MI355X only'):
        window_attention(torch.zeros(1, 49, 288), torch.zeros(288), torch.zeros(169, 3), 7, 7, 3, 0, 0.17)


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'gedepth_hip.h')).read()
    declared = set(re.findall(r'\b(ge_[a-z0-9_]+)\s*\(', header))
    assert declared == set(hip.SIGNATURES), declared ^ set(hip.SIGNATURES)
    if not hip.is_built():
        pytest.fail(f'{hip.LIB_PATH} missing: run gedepth_amd/csrc/build.sh')
    lib = ctypes.CDLL(hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert hip.lib().ge_abi_version() == 7
    # argument validation happens before any launch, so these are safe without a GPU
    assert hip.lib().ge_bilinear_fwd(None, None, 1, 1, 4, 4, 8, 8, 0, 0, None) == 10001
    assert hip.lib().ge_window_attn_bwd_workspace(2, 11, 35, 3) > 0


def test_msda_mm_location_division_is_ieee_division(tmp_path):
    """csrc/msda_mm.hip forms `offset / W` as q0 = off * RN(1/W), q = fma(fma(-q0, W, off), RN(1/W), q0).  The kink decisions of
    floor() in the sampling kernels hang on the last bit of that quotient, so the claim "bit-identical to IEEE division" is checked
    exhaustively on the host: every bf16 offset x every map width the kernel accepts (tools/ubench/divcheck.c, plain C with fmaf)."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('gcc not available')
    exe = str(tmp_path / 'divcheck')
    subprocess.run(['gcc', '-O2', '-ffp-contract=off', os.path.join(ROOT, 'tools', 'ubench', 'divcheck.c'), '-o', exe, '-lm'], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert 'mismatches 0' in out, out


def test_official_swin_checkpoint_loads_like_the_reference(golden):
    """§8 f2: ``DepthFormerSwin.load_pretrained`` on an official-layout Swin checkpoint (5x5-window bias tables, 3-channel
    patch embedding) gives the tensors the reference's ``init_weights`` produced from the same file
    (tests/golden/make_golden_ckpt.py): key renames, unfold-order fix, bicubic table resize, zero 4th input channel."""
    import importlib.util
    import os.path as osp
    import tempfile
    spec = importlib.util.spec_from_file_location('make_golden_ckpt', osp.join(osp.dirname(__file__), 'golden', 'make_golden_ckpt.py'))
    gen_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen_mod)
    g = golden('swin_official_ckpt')
    from gedepth_amd.depth.models.builder import BACKBONES
    from gedepth_amd.mmrt.registry import build_from_cfg
    cfg = dict(type='DepthFormerSwin', pretrain_img_size=224, patch_size=4, window_size=7, mlp_ratio=4, strides=(4, 2, 2, 2),
               out_indices=(0, 1, 2, 3), qkv_bias=True, qk_scale=None, patch_norm=True, drop_rate=0., attn_drop_rate=0.,
               drop_path_rate=0.0, use_abs_pos_embed=False, act_cfg=dict(type='GELU'), norm_cfg=dict(type='LN', requires_grad=True),
               pretrain_style='official', conv_norm_cfg=dict(type='BN', requires_grad=True), depth=50, num_stages=0, USEPE=True,
               **gen_mod.ARCH)
    m = build_from_cfg(cfg, BACKBONES)
    with tempfile.TemporaryDirectory() as tmp:
        path = osp.join(tmp, 'swin_official.pth')
        torch.save({'model': gen_mod.fake_official_swin()}, path)
        m.load_pretrained(path)
    sd = m.state_dict()
    assert sorted(sd.keys()) == [str(k) for k in g['all_keys']]
    checked = 0
    for k in g.files:
        if k.startswith('key::'):
            ref = torch.from_numpy(g[k])
            assert torch.allclose(sd[k[5:]].float(), ref.float(), rtol=1e-6, atol=1e-6), k
            checked += 1
    assert checked >= 20
    w = sd['patch_embed.projection.weight']
    assert w.shape[1] == 4 and float(w[:, 3].abs().max()) == 0.0


def test_tuning_tables_are_lookup_only(monkeypatch, tmp_path):
    """The committed library-selection tables: the GEMM table parses (validators + >= 60 shapes, incl. the batched split-K
    weight gradients), the MIOpen find-db is seeded into a content-keyed scratch copy, and 'off' disables TunableOp."""
    import csv
    from gedepth_amd.mmrt import tuning
    rows = list(csv.reader(open(tuning.GEMM_TABLE)))
    assert any(r[0] == 'Validator' and r[1] == 'GCN_ARCH_NAME' and r[2].startswith('gfx950') for r in rows)
    shapes = [r for r in rows if r[0] != 'Validator']
    assert len(shapes) >= 60 and any('StridedBatched' in r[0] for r in shapes)
    monkeypatch.delenv('MIOPEN_USER_DB_PATH', raising=False)
    monkeypatch.setenv('XDG_CACHE_HOME', str(tmp_path))
    try:
        assert tuning.use_miopen_find_db() is True
        dst = os.environ['MIOPEN_USER_DB_PATH']
        files = sorted(f for f in os.listdir(tuning.MIOPEN_DB) if f.endswith('db.txt'))
        assert dst.startswith(str(tmp_path)) and sorted(os.listdir(dst)) == files
        # a private (0700, user-owned) directory, never the shared tempdir; stale / foreign content is always rewritten
        assert os.stat(dst).st_mode & 0o777 == 0o700 and os.stat(os.path.dirname(dst)).st_mode & 0o777 == 0o700
        with open(os.path.join(dst, files[0]), 'w') as fh:
            fh.write('tampered')
        os.environ.pop('MIOPEN_USER_DB_PATH')
        assert tuning.use_miopen_find_db() is True
        assert open(os.path.join(dst, files[0]), 'rb').read() == open(os.path.join(tuning.MIOPEN_DB, files[0]), 'rb').read()
        os.environ.pop('MIOPEN_USER_DB_PATH')
        os.rename(dst, dst + '.real')
        os.symlink(dst + '.real', dst)
        with pytest.raises(RuntimeError, match='symlink'):
            tuning.use_miopen_find_db()
    finally:
        os.environ.pop('MIOPEN_USER_DB_PATH', None)
    assert tuning.use_tuned_gemms('off') is False
    with pytest.raises(ValueError):
        tuning.use_tuned_gemms('sometimes')


def test_drop_path_bank_serves_every_layer_once_per_draw():
    """mmrt.bricks._DropPathBank (host logic of the fused DropPath residual): layers register on first use, every refill draws one
    uniform per (layer, sample), a layer asking twice triggers a refill, scales are 0 or 1 / keep and the keep rate follows 1 - p."""
    import torch
    from gedepth_amd.mmrt.bricks import DropPath, _DropPathBank
    bank = _DropPathBank()
    layers = [DropPath(p) for p in (0.1, 0.5, 0.9)]
    dev = torch.device('cpu')
    torch.manual_seed(3)
    for l in layers:                                                     # registration pass (every new layer forces a refill)
        bank.scale(l, 6, torch.float32, dev)
    assert len(bank.layers) == 3
    refills, last = 0, None
    seen = {i: [] for i in range(3)}
    for _ in range(5):                                                   # five "steps": layers run in order, each asks once per step
        for i, l in enumerate(layers):
            r = bank.scale(l, 6, torch.float32, dev)
            keep = 1 - l.drop_prob
            assert r.shape == (6,) and bool(((r == 0) | ((r - 1 / keep).abs() < 1e-6)).all()) and bank.used[i]
            if bank.rows is not last:
                refills, last = refills + 1, bank.rows
            seen[i].append(r.clone())
    assert refills <= 6                                                  # one draw per step serves all layers (not one per layer)
    assert any(not torch.equal(a, b) for a, b in zip(seen[0][:-1], seen[0][1:]))     # a layer never sees the same draw twice in a row
    kept = torch.zeros(3)
    n = 400
    for _ in range(n):
        for i, l in enumerate(layers):
            kept[i] += (bank.scale(l, 6, torch.float32, dev) > 0).float().mean()
    for i, l in enumerate(layers):
        assert abs(kept[i].item() / n - (1 - l.drop_prob)) < 0.06, (i, kept[i].item() / n)
    late = DropPath(0.2)                                                 # a layer that shows up later joins the bank
    assert bank.scale(late, 6, torch.float32, dev).shape == (6,) and len(bank.layers) == 4
    assert bank.scale(layers[1], 4, torch.float32, dev).shape == (4,)    # another batch size: refill


def test_gemm_policy_table():
    """kernels.gemm_own: the shapes of the bench workload that go to ge_gemm_nt, and the refusals (host logic, no GPU)."""
    from gedepth_amd import kernels
    assert kernels.gemm_own(197120, 96, 288) and kernels.gemm_own(197120, 288, 96) and kernels.gemm_own(49280, 192, 192)
    assert not kernels.gemm_own(788480, 512, 768)        # the library solution is faster there
    assert not kernels.gemm_own(3080, 3072, 768)         # 39 tiles: no split-K in the kernel
    assert not kernels.gemm_own(197120, 96, 292) and not kernels.gemm_own(0, 96, 288)
    for (k, n), (lo, hi) in kernels.GEMM_OWN.items():
        assert k % 8 == 0 and n % 8 == 0 and 0 < lo < hi
