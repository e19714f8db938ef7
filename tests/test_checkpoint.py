"""quantized_checkpoint.pth.tar (quant_train.py:665-670): writer / loader of the reference's format.
CPU: structure round trip, strictness, and - in the build container - the file the LIVE reference writes.
GPU: a network restored from the checkpoint alone (other float weights, no calibration) gives the same logits."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GROUPS = ("convbn_scaling_factor", "fc_scaling_factor", "weight_integer", "bias_integer", "act_scaling_factor")


def _random_fill(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, b in model.named_buffers():
            if any(t in k for t in GROUPS):
                b.copy_(torch.randint(-100, 100, b.shape, generator=g).to(b.dtype))


def test_checkpoint_structure_round_trip(tmp_path):
    from hawq_amd.api import build_quantized_resnet, load_quantized_checkpoint, save_quantized_checkpoint
    a = build_quantized_resnet("resnet18", "uniform8", seed=1)
    _random_fill(a, 3)
    path = str(tmp_path / "quantized_checkpoint.pth.tar")
    save_quantized_checkpoint(a, path)
    ck = torch.load(path, map_location="cpu")
    assert tuple(ck) == GROUPS  # the reference's five dicts, in its order
    sd = a.state_dict()
    for g in GROUPS:
        assert set(ck[g]) == {k for k in sd if g in k} and ck[g]
    b = build_quantized_resnet("resnet18", "uniform8", seed=2)
    load_quantized_checkpoint(b, path)
    assert b.is_frozen() and b.engine_defaults == {"from_buffers": True}
    for k, v in a.state_dict().items():
        if any(g in k for g in GROUPS):
            assert torch.equal(b.state_dict()[k], v), k
    # a checkpoint of another architecture is refused, not half-loaded
    c = build_quantized_resnet("resnet50", "uniform8", seed=2)
    with pytest.raises(KeyError):
        load_quantized_checkpoint(c, path)
    with pytest.raises(KeyError):
        load_quantized_checkpoint(b, {"weight_integer": {}})


def test_dataparallel_prefixed_checkpoint_and_partial_files(tmp_path):
    """validate() saves the state_dict of the DataParallel-wrapped network (quant_train.py:358, 665-670): keys carry a
    'module.' prefix.  They must load like un-prefixed ones; a file that lacks tensors the integer engine needs is
    refused even with strict=False (it must never run on placeholder buffers)."""
    from hawq_amd.api import build_quantized_resnet, load_checkpoint, load_quantized_checkpoint
    a = build_quantized_resnet("resnet18", "uniform8", seed=1)
    _random_fill(a, 4)
    sd = a.state_dict()
    ck = {g: {"module." + k: v.clone() for k, v in sd.items() if g in k} for g in GROUPS}
    b = build_quantized_resnet("resnet18", "uniform8", seed=2)
    load_quantized_checkpoint(b, ck, strict=True)
    assert b.engine_defaults == {"from_buffers": True}
    for k, v in sd.items():
        if any(g in k for g in GROUPS):
            assert torch.equal(b.state_dict()[k], v), k
    # drop one conv's integer weights: refused, strict or not
    victim = next(k for k in ck["weight_integer"] if "stage2" in k)
    del ck["weight_integer"][victim]
    for strict in (True, False):
        c = build_quantized_resnet("resnet18", "uniform8", seed=2)
        with pytest.raises(KeyError):
            load_quantized_checkpoint(c, ck, strict=strict)
        assert not getattr(c, "engine_defaults", {}).get("from_buffers", False)
    # new float parameters make the integer buffers stale: a later float load stops trusting them
    fl = {"state_dict": {"module." + k: v for k, v in a.state_dict().items()}}
    load_checkpoint(b, fl)
    assert b.engine_defaults == {"from_buffers": False}
    # ... and a raw load_state_dict drops a cached plan
    b._engine = object()
    b.load_state_dict(a.state_dict(), strict=False)
    assert b._engine is None


@pytest.mark.reference
def test_loads_the_file_the_live_reference_writes(tmp_path):
    """The reference's own frozen ResNet18 -> its torch.save(...) of quant_train.py:665-670 -> our loader (strict):
    every key finds its buffer and the integer weights / biases / scales arrive unchanged."""
    from hawq_amd.api import build_quantized_resnet, load_quantized_checkpoint
    from hawq_amd.skeleton import synthetic_images
    from oracle import ref_live
    q = ref_live.build_reference_model("resnet18", "uniform8", seed=0)
    ref_live.calibrate_and_freeze(q, synthetic_images(2, seed=0))
    with torch.no_grad():
        q(synthetic_images(2, seed=0))  # a frozen forward fills weight_integer / bias_integer / scales
    sd = q.state_dict()
    path = str(tmp_path / "quantized_checkpoint.pth.tar")
    torch.save({g: {k: v for k, v in sd.items() if g in k} for g in GROUPS}, path)  # quant_train.py:665-670 verbatim
    ours = build_quantized_resnet("resnet18", "uniform8", seed=5)  # different float weights on purpose
    load_quantized_checkpoint(ours, path, strict=True)
    mine = ours.state_dict()
    n = 0
    for g in GROUPS:
        for k, v in sd.items():
            if g in k:
                assert torch.equal(mine[k].cpu().float().reshape(-1), v.cpu().float().reshape(-1)), k
                n += 1
    assert n > 60


@pytest.mark.reference
def test_loads_a_float_checkpoint_of_the_live_reference():
    """checkpoint.pth.tar as quant_train.py writes it ({'state_dict': DataParallel-prefixed}) from the reference's
    calibrated ResNet18 -> hawq_amd.api.load_checkpoint: every float weight, BN statistic and activation range arrives,
    nothing is missing or unexpected (same module / buffer names)."""
    from hawq_amd.api import build_quantized_resnet, load_checkpoint
    from hawq_amd.skeleton import synthetic_images
    from oracle import ref_live
    q = ref_live.build_reference_model("resnet18", "uniform8", seed=0)
    with torch.no_grad():
        q(synthetic_images(2, seed=0))  # un-frozen forward: QuantAct ranges initialise (quant_modules.py:247-250)
    sd = q.state_dict()
    ckpt = {"epoch": 1, "arch": "resnet18", "state_dict": {"module." + k: v for k, v in sd.items()}, "best_acc1": 0.0}
    ours = build_quantized_resnet("resnet18", "uniform8", seed=5)
    ours, missing, unexpected = load_checkpoint(ours, ckpt)
    assert not missing and not unexpected
    assert ours.is_frozen()
    mine = ours.state_dict()
    n = 0
    for k, v in sd.items():
        if any(t in k for t in ("num_batches_tracked", "weight_integer", "bias_integer")):
            continue
        assert torch.equal(mine[k].cpu().reshape(-1).float(), v.cpu().reshape(-1).float()), k
        n += 1
    assert n > 150


@pytest.mark.gpu
def test_float_checkpoint_round_trip_reproduces_logits():
    from hawq_amd.api import build_quantized_resnet, calibrate, load_checkpoint
    from hawq_amd.skeleton import synthetic_images
    x = synthetic_images(4, seed=3).cuda()
    a = build_quantized_resnet("resnet50", "bops_0.5", seed=0).cuda()
    calibrate(a, synthetic_images(8, seed=0).cuda())
    ya = a(x).clone()
    ckpt = {"state_dict": {"module." + k: v.detach().cpu() for k, v in a.state_dict().items()}}
    b = build_quantized_resnet("resnet50", "bops_0.5", seed=9).cuda()
    b, missing, unexpected = load_checkpoint(b, ckpt)
    assert not missing and not unexpected
    assert torch.equal(b(x), ya)


@pytest.mark.gpu
@pytest.mark.parametrize("arch,scheme", [("resnet18", "uniform8"), ("resnet50", "uniform4")])
def test_network_restored_from_checkpoint_alone(tmp_path, arch, scheme):
    from hawq_amd.api import build_quantized_resnet, calibrate, load_quantized_checkpoint, save_quantized_checkpoint
    from hawq_amd.skeleton import synthetic_images
    x = synthetic_images(4, seed=3).cuda()
    a = build_quantized_resnet(arch, scheme, seed=0).cuda()
    calibrate(a, synthetic_images(8, seed=0).cuda())
    ya = a(x).clone()
    path = str(tmp_path / "quantized_checkpoint.pth.tar")
    save_quantized_checkpoint(a, path)
    b = build_quantized_resnet(arch, scheme, seed=9).cuda()  # unrelated float weights, never calibrated
    load_quantized_checkpoint(b, path)
    yb = b(x)
    assert b._engine is not None and b._engine.from_buffers
    assert torch.equal(ya, yb)
    with torch.no_grad():   # the module-by-module path runs on the loaded integers and activation scales too
        assert torch.equal(b.forward_modules(x), a.forward_modules(x))
    # ... and from the bit-packed archive of the same checkpoint (hawq_amd.bitpack)
    from hawq_amd import bitpack
    packed = str(tmp_path / "packed.pth.tar")
    bitpack.pack_quantized_checkpoint(path, packed)
    c = build_quantized_resnet(arch, scheme, seed=11).cuda()
    bitpack.load_packed_checkpoint(c, packed)
    assert torch.equal(c(x), ya)


@pytest.mark.gpu
def test_mobilenetv2_restored_from_checkpoint_alone(tmp_path):
    """Q_MobileNetV2 save -> load into a network with unrelated float weights: the fused plan AND the module path must run on the
    loaded integers (a model without the ResNet engine silently re-quantised its synthetic floats).  The reference's five
    checkpoint groups omit QuantConv2d's scale buffer (the classifier): our file carries it in a sixth group, and a file without it
    is refused instead of running on a placeholder scale."""
    from hawq_amd.api import build_quantized_model, calibrate, load_quantized_checkpoint, save_quantized_checkpoint
    from hawq_amd.skeleton import synthetic_images
    x = synthetic_images(4, seed=3).cuda()
    a = build_quantized_model("mobilenetv2_w1", "uniform8", seed=0).cuda()
    calibrate(a, synthetic_images(8, seed=0).cuda())
    ya = a(x).clone()
    with torch.no_grad():
        ya_mod = a.forward_modules(x).clone()
    path = str(tmp_path / "quantized_checkpoint.pth.tar")
    save_quantized_checkpoint(a, path)
    ck = torch.load(path, map_location="cpu")
    assert list(ck["conv_scaling_factor"]) == ["output.conv_scaling_factor"]
    b = build_quantized_model("mobilenetv2_w1", "uniform8", seed=9).cuda()   # unrelated float weights, never calibrated
    load_quantized_checkpoint(b, path)
    assert torch.equal(b(x), ya) and b._engine is not None and b._engine.from_buffers
    with torch.no_grad():
        assert torch.equal(b.forward_modules(x), ya_mod)
    del ck["conv_scaling_factor"]      # what the reference's validate() would have written
    c = build_quantized_model("mobilenetv2_w1", "uniform8", seed=9).cuda()
    with pytest.raises(KeyError):
        load_quantized_checkpoint(c, ck)


def test_bitpack_round_trip_and_size(tmp_path):
    """hawq_amd.bitpack (README.md:61 of the reference: BitPack on weight_integer / quantized_checkpoint.pth.tar): field
    packing round-trips for every width, a packed W4 / W8 checkpoint restores the very same buffers and is 6-8x / ~4x
    smaller than the fp32 file."""
    import os
    from hawq_amd import bitpack
    from hawq_amd.api import build_quantized_resnet, load_quantized_checkpoint, save_quantized_checkpoint
    rng = np.random.default_rng(0)
    for bits in range(2, 10):
        a = rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), (3, 5, 7))
        s = bitpack.pack_tensor(a, bits)
        assert s.size == (a.size * bits + 7) // 8
        assert np.array_equal(bitpack.unpack_tensor(s, bits, a.shape), a) and bitpack.needed_bits(a) <= bits
    assert bitpack.needed_bits(np.array([-8, 7])) == 4 and bitpack.needed_bits(np.array([-9])) == 5 and bitpack.needed_bits(np.array([127.0])) == 8
    with pytest.raises(ValueError):
        bitpack.pack_tensor(np.array([8]), 4)
    for scheme, lo, hi, ratio in (("uniform4", -8, 8, 5.0), ("uniform8", -128, 128, 3.5)):
        a = build_quantized_resnet("resnet18", scheme, seed=1)
        g = torch.Generator().manual_seed(5)
        with torch.no_grad():
            for k, b in a.named_buffers():
                if "weight_integer" in k:
                    wlo, whi = (lo, hi) if "quant_init" not in k and "quant_output" not in k else (-128, 128)
                    b.copy_(torch.randint(wlo, whi, b.shape, generator=g).to(b.dtype))
                elif any(t in k for t in GROUPS):
                    b.copy_(torch.randint(-100, 100, b.shape, generator=g).to(b.dtype))
        plain, packed = str(tmp_path / f"q_{scheme}.pth.tar"), str(tmp_path / f"p_{scheme}.pth.tar")
        save_quantized_checkpoint(a, plain)
        bitpack.pack_quantized_checkpoint(plain, packed)
        assert os.path.getsize(plain) / os.path.getsize(packed) > ratio, (os.path.getsize(plain), os.path.getsize(packed))
        b = build_quantized_resnet("resnet18", scheme, seed=2)
        bitpack.load_packed_checkpoint(b, packed)
        assert b.engine_defaults == {"from_buffers": True}
        for k, v in a.state_dict().items():
            if any(t in k for t in GROUPS):
                assert torch.equal(b.state_dict()[k], v), k
    with pytest.raises(KeyError):
        bitpack.unpack_quantized_checkpoint({"weight_integer": {}})
