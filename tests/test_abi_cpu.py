"""CPU tests (-m "not gpu"): the C-ABI library loads, exports every symbol include/demonet_hip.h declares, the ctypes
mirrors match the C structs, and the host-side lowering (BN fold, weight packing, graph IR) is self-consistent.
No compute calls: there is no GPU here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from demonet_amd import _lib, models, spec, synth
from demonet_amd.plan import LoweredModel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "demonet_hip.h")


def _declared_symbols():
    txt = open(HEADER).read()
    return sorted(set(re.findall(r"DN_API\s+[\w\s\*]+?\b(dn_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        from demonet_amd import build
        build.build(verbose=False)
    L = C.CDLL(_lib.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(L, s), f"{s} declared in demonet_hip.h but not exported"
    assert set(syms) == set(_lib.EXPORTS), "ctypes binding and header disagree"
    L.dn_abi_version.restype = C.c_int
    assert L.dn_abi_version() == _lib.DN_ABI_VERSION


def test_library_exports_nothing_the_headers_do_not_declare():
    """Every dn_* symbol the product library exports is declared in include/demonet_hip.h (the boundary) or include/demonet_hip_debug.h
    (the test-support entry points); the probe hooks (dn_debug_*_stamps, dn_debug_pw_tile) exist in the dev build only."""
    if not os.path.exists(_lib.LIB_PATH):
        from demonet_amd import build
        build.build(verbose=False)
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split()[-1].startswith("dn_")})
    dbg = open(os.path.join(ROOT, "include", "demonet_hip_debug.h")).read()
    declared = set(_declared_symbols()) | set(re.findall(r"DN_API\s+[\w\s\*]+?\b(dn_\w+)\s*\(", dbg))
    assert set(exported) == declared, sorted(set(exported) ^ declared)
    assert sorted(s for s in exported if s.startswith("dn_debug_")) == ["dn_debug_clear_graphs", "dn_debug_head_fused_launches", "dn_debug_head_softmax_launches"]


def test_struct_layout_matches_c(tmp_path):
    """sizeof/offsetof of the ctypes mirrors against the real header (compiled with gcc)."""
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "demonet_hip.h"\n'
                   'int main(){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(dn_tensor_desc), sizeof(dn_op_desc), '
                   'sizeof(dn_model_desc), offsetof(dn_op_desc, w_off), offsetof(dn_model_desc, anchors), '
                   'offsetof(dn_model_desc, score_thresh));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    a = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert a[0] == C.sizeof(_lib.TensorDesc)
    assert a[1] == C.sizeof(_lib.OpDesc)
    assert a[2] == C.sizeof(_lib.ModelDesc)
    assert a[3] == _lib.OpDesc.w_off.offset
    assert a[4] == _lib.ModelDesc.anchors.offset
    assert a[5] == _lib.ModelDesc.score_thresh.offset


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libdemonet_hip.so")
    with pytest.raises(_lib.HipLibraryMissing):
        _lib.lib()


@pytest.mark.parametrize("name,ncls,nparams", [("ssdlite320_mobilenet_v3_large", 91, 476)])
def test_state_dict_keys_and_lowering(name, ncls, nparams):
    m = getattr(models, name)(num_classes=ncls)
    sd = m.state_dict()
    assert len(sd) == nparams                                   # 476 entries for SSDLite-V3
    assert "backbone.features.0.0.0.weight" in sd and "head.classification_head.module_list.0.1.bias" in sd
    models.load_synthetic(m, 0)
    lm = LoweredModel(m.graph, m.state_dict())
    assert lm.desc.num_anchors == 3234 and lm.desc.n_ops == len(m.graph.nodes)
    # BN fold check on the first pointwise op against an explicit fp64 computation
    nd = next(n for n in m.graph.nodes if n.op == "pw")
    o = next(o for o, n in zip(lm.ops, m.graph.nodes) if n is nd)
    w = np.frombuffer(lm.blob, dtype=np.float16, count=nd.cout * nd.cin, offset=o.w_off).reshape(nd.cout, nd.cin)
    s = synth.state_dict(m.graph, 0)
    g, b = s[nd.bn_key + ".weight"].astype(np.float64), s[nd.bn_key + ".bias"].astype(np.float64)
    mu, var = s[nd.bn_key + ".running_mean"].astype(np.float64), s[nd.bn_key + ".running_var"].astype(np.float64)
    sc = g / np.sqrt(var + nd.bn_eps)
    ref = s[nd.conv_key + ".weight"].reshape(nd.cout, nd.cin).astype(np.float64) * sc[:, None]
    np.testing.assert_allclose(w.astype(np.float64), ref, rtol=1e-3, atol=1e-4)
    bias = np.frombuffer(lm.blob, dtype=np.float32, count=nd.cout, offset=o.b_off)
    np.testing.assert_allclose(bias, b - mu * sc, rtol=1e-6, atol=1e-6)


def test_checkpoint_ingestion_by_reference_key_names(tmp_path):
    """a torchvision-format checkpoint (a pickled state_dict with the reference's key names, BN buffers and
    num_batches_tracked included; ssd_mobilenetv3.py:20-23,222-224) loads through the factory's load_state_dict, strictly, and
    lowers to exactly the blob the same weights give when injected directly."""
    m = models.ssdlite320_mobilenet_v3_large(num_classes=91)
    src = {k: torch.from_numpy(np.asarray(v).copy()) for k, v in synth.state_dict(m.graph, 2).items()}
    assert any(k.endswith("num_batches_tracked") for k in src)
    path = tmp_path / "ssdlite320_mobilenet_v3_large_coco-synthetic.pth"
    torch.save(src, path)
    ck = torch.load(path, map_location="cpu")
    res = m.load_state_dict(ck)                                    # strict=True: any key mismatch raises
    assert not res.missing_keys and not res.unexpected_keys
    a = LoweredModel(m.graph, m.state_dict())
    b = LoweredModel(m.graph, {k: v.numpy() for k, v in src.items()})
    assert bytes(a.blob) == bytes(b.blob)
    # a checkpoint of another architecture / class count is refused, as nn.Module.load_state_dict does for the reference
    bad = dict(ck)
    bad.pop("head.classification_head.module_list.0.1.bias")
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad)
    with pytest.raises(RuntimeError):
        models.ssdlite320_mobilenet_v3_large(num_classes=21).load_state_dict(ck)      # 6*21 != 6*91 output channels


def test_factory_api_mirrors_reference():
    with pytest.warns(UserWarning):
        m = models.ssdlite320_mobilenet_v3_large(num_classes=5, size=(300, 300), score_thresh=0.3)     # ssd_mobilenetv3.py:183-184
    assert m.score_thresh == 0.3 and m.nms_thresh == 0.55 and m.detections_per_img == 300 and m.topk_candidates == 300
    assert m.graph.num_classes == 5 and not m.training
    with pytest.raises(RuntimeError):
        models.ssdlite320_mobilenet_v3_large(pretrained=True)
    v = models.ssd300_vgg16(num_classes=91)
    assert v.graph.num_anchors() == 8732 and v.score_thresh == 0.01 and v.topk_candidates == 400
    h = models.ssd_lite_mobilenet_v2(num_classes=21)
    assert h.score_thresh == 0.5 and h.graph.num_anchors() == 3234
    assert models.__dict__["ssdlite320_mobilenet_v3_large"] is models.ssdlite320_mobilenet_v3_large   # train.py:154 lookup
    m.train()
    with pytest.raises(ValueError):
        m([torch.zeros(3, 8, 8)])
    m.eval()
    with pytest.raises(RuntimeError):
        m([torch.zeros(3, 320, 320)])          # CPU input: the product path has no fallback
    from demonet_amd.pipeline import ForwardPipeline
    from demonet_amd import engine
    with pytest.raises(RuntimeError):
        ForwardPipeline(m, 4, device="cpu")    # ... nor does the pipeline of forwards in flight
    with pytest.raises(ValueError):
        ForwardPipeline(m, 4, depth=0)
    with pytest.raises(RuntimeError):
        engine.evaluate(m, [([torch.zeros(3, 320, 320)], [{"image_id": 1}])], device="cpu")


def test_graph_geometry_matches_survey():
    g = spec.ssdlite320_mobilenet_v3_large_graph(91)
    assert [(g.t(f).c, g.t(f).h) for f in g.features] == [(672, 20), (480, 10), (512, 5), (256, 3), (256, 2), (128, 1)]
    macs = sum(g.t(n.out).h * g.t(n.out).w * n.cout * (n.cin if n.op == "pw" else n.k * n.k * (3 if n.op == "stem" else 1))
               for n in g.nodes if n.op in ("pw", "dw", "stem"))
    assert abs(macs / 1e6 - 583.17) < 1.0                       # 583.17 MMAC / image
    v = spec.ssd512_vgg16_graph(91)
    assert v.num_anchors() == 24732 and [g_.h for g_ in (v.t(f) for f in v.features)] == [64, 32, 16, 8, 6, 4, 1]


@pytest.mark.parametrize("name,factory,kw", [
    ("ssdlite320_mobilenet_v3_large", "ssdlite320_mobilenet_v3_large", {}),
    ("ssd_lite_mobilenet_v2", "ssd_lite_mobilenet_v2", {}),
    ("ssd300_vgg16", "ssd300_vgg16", {}),
    ("ssd512_vgg16", "ssd512_vgg16", {}),
])
def test_product_anchors_bit_equal_reference(golden_dir, name, factory, kw):
    """demonet_amd.anchors.default_boxes (the table the plan uploads) against the anchors the real reference's
    DefaultBoxGenerator produced (anchor_utils.py:75-126; goldens from tests/golden/make_golden.py): bit-equal."""
    import os
    from demonet_amd import anchors
    p = os.path.join(golden_dir, name + ".npz")
    if not os.path.exists(p):
        pytest.skip("no golden for " + name)
    z = np.load(p)
    g = getattr(models, factory)(num_classes=int(z["num_classes"]), **kw).graph
    grid = [(g.t(f).h, g.t(f).w) for f in g.features]
    a = anchors.default_boxes(grid, (g.size[1], g.size[0]), **g.anchor_spec)
    assert a.dtype == np.float32 and a.shape == z["anchors"].shape
    assert np.array_equal(a, z["anchors"])


def test_factory_kwargs_over_defaults():
    """{**defaults, **kwargs} of the reference factories (ssd_mobilenetv3.py:207-218, ssd_vgg16.py:200-206): image_mean / image_std
    reach the plan's dn_model_desc; the training-only SSD.__init__ arguments are accepted; anything else raises instead of being
    dropped; norm_layer's BatchNorm eps is honoured."""
    import functools
    from torch import nn
    m = models.ssdlite320_mobilenet_v3_large(num_classes=3, image_mean=[0.4, 0.5, 0.6], image_std=(0.2, 0.3, 0.4), iou_thresh=0.4,
                                             positive_fraction=0.3)
    assert m.graph.image_mean == [0.4, 0.5, 0.6] and m.graph.image_std == [0.2, 0.3, 0.4]
    low = LoweredModel(m.graph, {k: v.numpy() for k, v in m.state_dict().items()})
    assert [round(x, 6) for x in low.desc.mean] == [0.4, 0.5, 0.6] and [round(x, 6) for x in low.desc.std] == [0.2, 0.3, 0.4]
    d = models.ssdlite320_mobilenet_v3_large(num_classes=3)
    assert d.graph.image_mean == [0.5] * 3 and d.graph.image_std == [0.5] * 3
    v = models.ssd300_vgg16(num_classes=3, image_mean=[0.1, 0.2, 0.3])
    assert v.graph.image_mean == [0.1, 0.2, 0.3] and abs(v.graph.image_std[0] - 1 / 255.0) < 1e-12
    with pytest.raises(TypeError):
        models.ssdlite320_mobilenet_v3_large(width_mult=0.5)
    with pytest.raises(TypeError):
        models.ssd300_vgg16(nonsense=1)
    with pytest.raises(ValueError):
        models.ssd300_vgg16(image_std=[1.0, 2.0])
    with pytest.raises(NotImplementedError):
        models.ssdlite320_mobilenet_v3_large(norm_layer=nn.GroupNorm)
    # eps of the norm layer changes the folded weights
    sd = {k: v.numpy() for k, v in d.state_dict().items()}
    e = models.ssdlite320_mobilenet_v3_large(num_classes=3, norm_layer=functools.partial(nn.BatchNorm2d, eps=0.1, momentum=0.03))
    a, b = LoweredModel(d.graph, sd), LoweredModel(e.graph, sd)
    assert bytes(a.blob) != bytes(b.blob)


def test_plan_signature_tracks_weight_updates():
    """The cached device plan is keyed on identity / storage / version of every tensor and dropped by load_state_dict and
    _apply (.to(), .half() ...); `.data` edits need invalidate()."""
    m = models.ssdlite320_mobilenet_v3_large(num_classes=3)
    s0 = m._weights_signature()
    p = next(m.parameters())
    p.data.mul_(1.0)
    assert m._weights_signature() == s0                     # bypasses the version counter: documented, needs invalidate()
    with torch.no_grad():
        p.mul_(1.0)
    assert m._weights_signature() != s0
    s1 = m._weights_signature()
    m.load_state_dict({k: v.clone() for k, v in m.state_dict().items()}, assign=True)
    assert m._weights_signature() != s1
    m._handle = None
    m.invalidate()
    assert m._sig is None


def test_bench_fused_block_bytes_count_the_residual_once():
    """bench.py's roofline bytes of a fused inverted-residual launch: input once, output once, weights -- the residual of a block IS its
    input and must not be added again (round-4 review: the expdw family was over-credited by 21 %)."""
    import importlib.util
    sp = importlib.util.spec_from_file_location("bench_for_costs", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(bench)
    g = models.ssdlite320_mobilenet_v3_large(num_classes=91).graph
    n = 64
    costs = bench.op_costs(g, n)
    # block b3 (features.0.3): pw 24->72, dw 72 k3 s1, pw 72->24 + residual, all on the 80 x 80 map
    i = next(i for i, nd in enumerate(g.nodes) if nd.op == "pw" and nd.cin == 24 and nd.cout == 72 and g.t(nd.inp).h == 80 and g.nodes[i + 2].residual == nd.inp)
    mem = [i, i + 1, i + 2]
    m = n * 80 * 80
    want = 2 * m * 24 + 2 * m * 24 + (2 * 24 * 72 + 4 * 72) + (2 * 9 * 72 + 4 * 72) + (2 * 72 * 24 + 4 * 24)
    assert bench.fused_external_bytes(g, costs, mem, False) == want
    # a projection whose residual comes from elsewhere (not this launch's input) still pays for it
    assert bench.fused_external_bytes(g, costs, [i + 2], False) == 2 * m * 72 + 2 * m * 24 + (2 * 72 * 24 + 4 * 24) + 2 * m * 24


def test_bench_prices_the_post_process_by_what_the_library_launched():
    """Round-5 slip: bench.py read DN_HEAD_SOFTMAX with the wrong default and booked a whole-anchor softmax launch (156 MB in 21 us = "7.2 TB/s")
    for a launch that only touches the levels the fused head launch did not finish. The levels are now derived from the launched kernel's
    name, the small launch is priced with its own anchors, and the head launch with scores + boxes + histogram rows instead of logits."""
    import bench
    from demonet_amd import spec
    g = spec.GRAPHS["ssdlite320_mobilenet_v3_large"]()
    n, K, A = 64, g.num_classes, g.num_anchors()
    assert bench.softmax_levels(g, ["pw_direct_kernel<1,1>", "head_fused_kernel<5>"]) == ()
    sm = bench.softmax_levels(g, ["head_fused_kernel<5,softmax>"])
    assert sm == (0, 1)                                   # 20 x 20 and 10 x 10: >= 32 pixels per image
    full = bench.op_costs(g, n)[len(g.nodes)]
    part = bench.op_costs(g, n, sm)[len(g.nodes)]
    assert full["kernel"] == part["kernel"] == "softmax_decode_kernel"
    assert full["bytes"] == n * (A * (4 * K + 16 + 4 * (K - 1) + 16) + (A + 63) // 64 * 1024)
    left = A - (400 + 100) * 6
    assert left == 234 and part["bytes"] == n * (left * (4 * K + 16 + 4 * (K - 1) + 16) + 4 * 1024)
    assert part["bytes"] < 0.09 * full["bytes"]
    costs = bench.op_costs(g, n, sm)
    mem = [i for i, nd in enumerate(g.nodes) if nd.head]
    mem += [i for i, nd in enumerate(g.nodes) if nd.op == "dw" and any(g.nodes[j].inp == nd.out for j in mem)]
    plain = bench.fused_external_bytes(g, costs, sorted(mem), True)
    with_sm = bench.fused_external_bytes(g, costs, sorted(mem), True, sm)
    logits_large = 4.0 * n * (400 + 100) * 6 * K
    assert abs((plain - with_sm) - (logits_large / K - n * (14 + 5) * 1024)) < 1.0
