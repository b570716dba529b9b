"""CPU: host-side logic - module structure / checkpoint layout, the CPU plumbing run (BASELINE config 1),
the data-parallel helpers on gloo (world_size 2), oracle schedule invariants."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pfa_gan
from oracle import fill, nets, step as ostep

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gan-aug-pfa_amd")


def test_state_dict_manifest_matches_reference_layout():
    """Appendix A of SURVEY.md: key order, shapes, dtypes of the default networks (70 / 22 entries)."""
    G, D = pfa_gan.UNetGenerator(3, 3), pfa_gan.NLayerDiscriminator(6)
    gs, ds = nets.g_state_shapes(), nets.d_state_shapes()
    assert len(gs) == 70 and len(ds) == 22
    for m, shapes in ((G, gs), (D, ds)):
        sd = m.state_dict()
        assert list(sd) == list(shapes)
        for k, v in sd.items():
            assert tuple(v.shape) == tuple(shapes[k])
            assert v.dtype == (torch.int64 if k.endswith("num_batches_tracked") else torch.float32)
    assert sum(p.numel() for p in G.parameters()) == 41_828_995
    assert sum(p.numel() for p in D.parameters()) == 2_768_705
    assert sum(p.numel() for p in pfa_gan.UNetGenerator(3, 3, num_downs=6).parameters()) == 29_244_035


def test_cpu_modules_follow_the_oracle(golden):
    """On a CPU tensor the drop-in modules execute plain PyTorch; same numbers as the reference goldens."""
    z = golden("nets")
    G = pfa_gan.UNetGenerator(3, 3, num_downs=5, ngf=8)
    G.load_state_dict({k: torch.from_numpy(v) for k, v in fill.fill_state(nets.g_state_shapes(3, 3, 5, 8), seed=1).items()})
    x = torch.from_numpy(fill.uniform("small.x", (2, 3, 32, 32), seed=int(z["small.g.seed"])))
    np.testing.assert_allclose(G(x).detach().numpy(), z["small.g.y"], atol=1e-5, rtol=1e-4)
    with pytest.raises(ValueError):
        pfa_gan.UNetGenerator(3, 3, num_downs=4)


def test_instance_norm_variant_keys():
    """norm_layer=nn.InstanceNorm2d flips the conv biases on (models.py:171-174): 28 / 10 keys."""
    import functools
    G = pfa_gan.UNetGenerator(3, 3, norm_layer=torch.nn.InstanceNorm2d)
    D = pfa_gan.NLayerDiscriminator(6, norm_layer=torch.nn.InstanceNorm2d)
    assert len(G.state_dict()) == 28 and len(D.state_dict()) == 10
    Ga = pfa_gan.UNetGenerator(3, 3, norm_layer=functools.partial(torch.nn.InstanceNorm2d, affine=True))
    assert len(Ga.state_dict()) == 50


def _write_pairs(root, n, size):
    from PIL import Image
    rng = np.random.RandomState(0)
    for i in range(n):
        city = f"city{i}"
        os.makedirs(os.path.join(root, "images", city, "pair"))
        os.makedirs(os.path.join(root, "labels", city, "cm"))
        for nm in ("img1", "img2"):
            Image.fromarray(rng.randint(0, 256, (size, size, 3), dtype=np.uint8)).save(
                os.path.join(root, "images", city, "pair", nm + ".png"))
        Image.fromarray((rng.rand(size, size) > 0.7).astype(np.uint8) * 255).save(
            os.path.join(root, "labels", city, "cm", "cm.png"))


def test_baseline_config1_cpu_plumbing(tmp_path):
    """BASELINE config 1: 4 synthetic 64x64 RGB pairs, batch 2, 1 epoch on PyTorch-CPU (num_downs=6: the default 7
    cannot run 64x64) -> checkpoints in the reference layout, consumed by the generation script."""
    root = str(tmp_path)
    _write_pairs(root, 4, 64)
    env = dict(os.environ, TQDM_DISABLE="1")
    cmd = [sys.executable, os.path.join(PKG, "train_gan.py"), "--root", root, "--images-dir", os.path.join(root, "images"),
           "--labels-dir", os.path.join(root, "labels"), "--device", "cpu", "--size", "64", "--num-downs", "6", "--ngf", "8",
           "--ndf", "8", "--batch-size", "2", "--epochs", "1", "--workers", "0", "--seed", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Avg Loss D" in r.stdout
    gpath = os.path.join(root, "gan_checkpoints", "generator_epoch_1.pth")
    sd = torch.load(gpath)
    assert list(sd) == list(nets.g_state_shapes(3, 3, 6, 8))
    assert int(sd["model.model.1.model.2.num_batches_tracked"]) == 4      # 2 steps x 2 generator forwards
    assert os.path.exists(os.path.join(root, "gan_checkpoints", "discriminator_epoch_1.pth"))
    assert any(f.startswith("sample_") for f in os.listdir(os.path.join(root, "gan_samples")))
    out = os.path.join(root, "synthetic_data")
    cmd = [sys.executable, os.path.join(PKG, "generate_synthetic_data.py"), "--images-dir", os.path.join(root, "images"),
           "--labels-dir", os.path.join(root, "labels"), "--checkpoint", gpath, "--out-dir", out, "--device", "cpu",
           "--size", "64", "--num-downs", "6", "--ngf", "8", "--workers", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    from pfa_gan.data import scan_dataset
    found = scan_dataset(os.path.join(out, "images"), os.path.join(out, "labels"), is_synthetic=True)
    assert len(found) == 4 and all(s["city"].endswith("_synth") for s in found)


def test_oracle_minimal_schedule_side_effects():
    """F5 / Appendix B: one step leaves G's BN counters at +2 and D's at +3."""
    gst = fill.fill_state(nets.g_state_shapes(3, 3, 5, 8), seed=5, bn_noise=False)
    dst = fill.fill_state(nets.d_state_shapes(6, 8, 3), seed=6, bn_noise=False)
    a, b = fill.uniform("sched.a", (2, 3, 32, 32)), fill.uniform("sched.b", (2, 3, 32, 32))
    ld, lg, _ = ostep.gan_step(gst, dst, ostep.new_adam_state(gst), ostep.new_adam_state(dst), a, b, 5, 3)
    assert int(gst["model.model.1.model.2.num_batches_tracked"]) == 2
    assert int(dst["model.3.num_batches_tracked"]) == 3
    assert 0.6 < ld < 0.8 and 40 < lg < 80


# ------------------------------------------------------------------------------ data parallel, gloo, 2 ranks
def _dp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                       # deliberately different initial weights per rank
        G = pfa_gan.UNetGenerator(3, 3, num_downs=5, ngf=8)
        D = pfa_gan.NLayerDiscriminator(6, ndf=8)
        pfa_gan.broadcast_model(G)
        pfa_gan.broadcast_model(D)
        og = torch.optim.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.999))
        od = torch.optim.Adam(D.parameters(), lr=1e-4, betas=(0.5, 0.999))
        batches = [dict(image1=torch.from_numpy(fill.uniform(f"dp.a{i}", (2, 3, 32, 32), seed=rank)),
                        image2=torch.from_numpy(fill.uniform(f"dp.b{i}", (2, 3, 32, 32), seed=rank))) for i in range(2)]
        # gradient averaging itself, on known values
        for p in G.parameters():
            p.grad = torch.full_like(p, float(rank + 1))
        pfa_gan.allreduce_mean_grads(G, dist.group.WORLD)
        assert all(torch.allclose(p.grad, torch.full_like(p, 1.5)) for p in G.parameters())
        ld, lg = pfa_gan.train_gan_one_epoch(G, D, batches, og, od, process_group=dist.group.WORLD)
        flat = torch.cat([p.detach().reshape(-1) for p in list(G.parameters()) + list(D.parameters())])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        if rank == 0:
            torch.save(dict(same=bool(torch.equal(gathered[0], gathered[1])), ld=ld, lg=lg, flat=flat,
                            bn=[b.clone() for b in G.buffers()]), out)
    finally:
        dist.destroy_process_group()


def test_data_parallel_two_ranks_gloo(tmp_path):
    """Two ranks, different shards: after every optimizer step both replicas hold identical parameters (rank 0's
    broadcast + averaged gradients), BN statistics stay per-replica, and the result is not the single-rank one."""
    out = str(tmp_path / "dp.pt")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_dp_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out, weights_only=False)
    assert r["same"], "replicas diverged"
    assert np.isfinite(r["ld"]) and np.isfinite(r["lg"])
    # single process on rank 0's shard only, same initial weights -> different result (the all-reduce mattered)
    torch.manual_seed(100)
    G = pfa_gan.UNetGenerator(3, 3, num_downs=5, ngf=8)
    D = pfa_gan.NLayerDiscriminator(6, ndf=8)
    og = torch.optim.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.999))
    od = torch.optim.Adam(D.parameters(), lr=1e-4, betas=(0.5, 0.999))
    batches = [dict(image1=torch.from_numpy(fill.uniform(f"dp.a{i}", (2, 3, 32, 32), seed=0)),
                    image2=torch.from_numpy(fill.uniform(f"dp.b{i}", (2, 3, 32, 32), seed=0))) for i in range(2)]
    pfa_gan.train_gan_one_epoch(G, D, batches, og, od)
    flat = torch.cat([p.detach().reshape(-1) for p in list(G.parameters()) + list(D.parameters())])
    assert not torch.equal(flat, r["flat"])
    # per-replica BN statistics: rank 0 saw rank 0's shard only -> its running buffers equal the single-rank ones
    # after the first forward would differ only through the weights; counters are exact
    assert [int(b) for b in r["bn"] if b.dtype == torch.int64] == [int(b) for b in G.buffers() if b.dtype == torch.int64]


def test_default_init_matches_reference_seed_for_seed(golden):
    """the reference never re-initialises (models.py:149-247 rely on reset_parameters), so a user who seeds
    torch and builds the nets gets the reference's parameters only if these classes create the same sub-modules in the
    same order: per-tensor sums of the seeded default state_dicts, and the key order, against tests/golden/init.npz
    (generated from the reference classes)."""
    from make_golden import INIT_CASES
    z = golden("init")
    for name, gkw, dkw, seed in INIT_CASES:
        torch.manual_seed(seed)
        G = pfa_gan.UNetGenerator(3, 3, **gkw)
        D = pfa_gan.NLayerDiscriminator(6, **dkw)
        for tag, m in (("g", G), ("d", D)):
            st = m.state_dict()
            assert list(st.keys()) == str(z[f"init.{name}.{tag}.keys"]).split("\n"), (name, tag)
            sums = np.array([float(v.double().sum()) for v in st.values()])
            asums = np.array([float(v.double().abs().sum()) for v in st.values()])
            np.testing.assert_array_equal(sums, z[f"init.{name}.{tag}.sums"])
            np.testing.assert_array_equal(asums, z[f"init.{name}.{tag}.abs_sums"])


def _dp_skip_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=60))
    try:
        torch.manual_seed(100)
        G = pfa_gan.UNetGenerator(3, 3, num_downs=5, ngf=8)
        D = pfa_gan.NLayerDiscriminator(6, ndf=8)
        og = torch.optim.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.999))
        od = torch.optim.Adam(D.parameters(), lr=1e-4, betas=(0.5, 0.999))
        batches = [dict(image1=torch.from_numpy(fill.uniform(f"sk.a{i}", (2, 3, 32, 32), seed=rank)),
                        image2=torch.from_numpy(fill.uniform(f"sk.b{i}", (2, 3, 32, 32), seed=rank))) for i in range(3)]
        if rank == 1:
            batches[1] = None           # what collate_fn_skip_none yields when every sample of the batch failed to load
        pfa_gan.train_gan_one_epoch(G, D, batches, og, od, process_group=dist.group.WORLD)
        steps = torch.tensor([og.state[next(iter(G.parameters()))]["step"].item()])
        allsteps = [torch.empty_like(steps) for _ in range(world)]
        dist.all_gather(allsteps, steps)
        flat = torch.cat([p.detach().reshape(-1) for p in list(G.parameters()) + list(D.parameters())])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        if rank == 0:
            torch.save(dict(steps=[int(t) for t in allsteps], same=bool(torch.equal(gathered[0], gathered[1]))), out)
    finally:
        dist.destroy_process_group()


def test_data_parallel_skips_a_missing_batch_on_every_rank(tmp_path):
    """One rank's batch collates to None (dataset.py:13-18): every rank skips that iteration instead of one rank entering
    the gradient all-reduces alone (a hang until the collective times out)."""
    out = str(tmp_path / "skip.pt")
    port = 31500 + os.getpid() % 2000
    mp.spawn(_dp_skip_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out, weights_only=False)
    assert r["steps"] == [2, 2], r
    assert r["same"], "replicas diverged"


def test_siamese_unet_dropin_matches_reference_golden(golden):
    """`from models import SiameseUNet` (reference train.py:16, evaluate.py:12) resolves to a plain-torch module with the
    reference's state_dict keys (in order), seeded default init and train / eval arithmetic (tests/golden/siamese.npz,
    generated from the reference by tests/make_golden.py)."""
    sys.path.insert(0, PKG)
    try:
        import models as dropin
    finally:
        sys.path.remove(PKG)
    z = golden("siamese")
    torch.manual_seed(7)
    net = dropin.SiameseUNet(3, 2)
    st0 = net.state_dict()
    assert list(st0.keys()) == str(z["siamese.keys"]).split("\n")
    sums = np.array([float(v.double().sum()) for v in st0.values()])
    np.testing.assert_allclose(sums, z["siamese.init_sums"], rtol=0, atol=0)       # same RNG consumption order
    shapes = {k: tuple(v.shape) for k, v in st0.items()}
    st = fill.fill_state(shapes, seed=3)
    for k in shapes:
        if k.endswith("running_mean"):
            st[k] = fill.uniform(k, shapes[k], -0.2, 0.2, 3)
        elif k.endswith("running_var"):
            st[k] = fill.uniform(k, shapes[k], 0.5, 1.5, 3)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()}, strict=True)
    x1 = torch.from_numpy(fill.uniform("siamese.x1", (2, 3, 32, 32)))
    x2 = torch.from_numpy(fill.uniform("siamese.x2", (2, 3, 32, 32)))
    net.eval()
    with torch.no_grad():
        np.testing.assert_allclose(net(x1, x2).numpy(), z["siamese.out_eval"], rtol=1e-5, atol=1e-5)
    net.train()
    out = net(x1, x2)
    np.testing.assert_allclose(out.detach().numpy(), z["siamese.out_train"], rtol=1e-5, atol=1e-5)
    out.square().mean().backward()
    np.testing.assert_allclose(net.conv_last.weight.grad.numpy(), z["siamese.grad_conv_last"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(net.dconv_down1[0].weight.grad.numpy(), z["siamese.grad_down1"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(net.state_dict()["bottleneck.1.running_mean"].numpy(), z["siamese.rm_after_train"],
                               rtol=1e-5, atol=1e-6)


def test_dataset_contract_pixel_values_labels_and_layout(tmp_path):
    """The batch-dict contract the step consumes (reference dataset.py:13-36 collate + JointToTensor, :136-159 resize +
    normalise, :229-237 bad samples, :248-270 directory layouts), pinned on hand-written PNGs:
    image = x/255*2-1 exactly, label = (L > 128) as int64, synthetic samples are named <city>_synth, a sample whose file
    goes bad after the scan comes back as None and is dropped by collate_fn_skip_none."""
    from PIL import Image
    from pfa_gan import data as D

    h, w = 4, 6
    rng = np.random.RandomState(5)
    px1 = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
    px1[0, 0], px1[0, 1], px1[0, 2] = (0, 0, 0), (255, 255, 255), (128, 127, 129)
    px2 = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
    lab = np.array([[0, 127, 128, 129, 255, 200]] * h, dtype=np.uint8)          # the threshold is strict: 128 -> 0, 129 -> 1
    images, labels = tmp_path / "images", tmp_path / "labels"

    def real_city(name, a=px1, b=px2, m=lab):
        (images / name / "pair").mkdir(parents=True)
        (labels / name / "cm").mkdir(parents=True)
        Image.fromarray(a).save(images / name / "pair" / "img1.png")
        Image.fromarray(b).save(images / name / "pair" / "img2.png")
        Image.fromarray(m, mode="L").save(labels / name / "cm" / "cm.png")

    real_city("paris")
    real_city("pisa")
    (images / "nolabel" / "pair").mkdir(parents=True)                            # label missing: skipped at the scan
    Image.fromarray(px1).save(images / "nolabel" / "pair" / "img1.png")
    Image.fromarray(px2).save(images / "nolabel" / "pair" / "img2.png")
    (images / "broken" / "pair").mkdir(parents=True)                            # unreadable image: skipped at the scan
    (labels / "broken" / "cm").mkdir(parents=True)
    (images / "broken" / "pair" / "img1.png").write_bytes(b"not a png")
    Image.fromarray(px2).save(images / "broken" / "pair" / "img2.png")
    Image.fromarray(lab, mode="L").save(labels / "broken" / "cm" / "cm.png")
    (images / "stray.txt").write_text("not a city directory")

    samples = D.scan_dataset(str(images), str(labels))
    assert [s["city"] for s in samples] == ["paris", "pisa"]
    assert samples[0]["img1"].endswith(os.path.join("paris", "pair", "img1.png"))
    assert samples[0]["label"].endswith(os.path.join("paris", "cm", "cm.png"))

    # no resize: exact values
    ds = D.BaseChangeDetectionDataset(samples, target_size=(h, w), augment=False)
    item = ds[0]
    want1 = torch.from_numpy(px1).permute(2, 0, 1).float().div(255) * 2.0 - 1.0      # TF.to_tensor, then JointNormalize
    want2 = torch.from_numpy(px2).permute(2, 0, 1).float().div(255) * 2.0 - 1.0
    assert item["image1"].dtype == torch.float32 and item["image1"].shape == (3, h, w)
    assert torch.equal(item["image1"], want1) and torch.equal(item["image2"], want2)
    assert float(item["image1"][0, 0, 0]) == -1.0 and float(item["image1"][0, 0, 1]) == 1.0
    assert item["label"].dtype == torch.int64 and item["label"].shape == (h, w)
    assert item["label"][0].tolist() == [0, 0, 0, 1, 1, 1]
    assert item["city"] == "paris"

    # resize path (train_gan.py:32,108 resizes everything to TARGET_SIZE): shape, dtype, range; labels stay {0, 1} (nearest)
    big = D.BaseChangeDetectionDataset(samples, target_size=(8, 8), augment=False)[1]
    assert big["image1"].shape == (3, 8, 8) and big["image1"].dtype == torch.float32
    assert float(big["image1"].min()) >= -1.0 and float(big["image1"].max()) <= 1.0
    assert big["label"].shape == (8, 8) and big["label"].dtype == torch.int64
    assert set(big["label"].unique().tolist()) <= {0, 1} and int(big["label"].sum()) > 0
    flat = np.full((h, w, 3), 77, dtype=np.uint8)                                # a constant image stays constant under bilinear
    real_city("flat", a=flat, b=flat)
    up = D.BaseChangeDetectionDataset(D.scan_dataset(str(images), str(labels)), target_size=(9, 5))
    flat_item = [up[i] for i in range(len(up)) if up.samples[i]["city"] == "flat"][0]
    assert torch.allclose(flat_item["image1"], torch.full((3, 9, 5), 77 / 255 * 2 - 1), atol=1e-6)

    # collate: dict batches, None samples dropped, all-None -> None
    batch = D.collate_fn_skip_none([ds[0], None, ds[1]])
    assert batch["image1"].shape == (2, 3, h, w) and batch["label"].shape == (2, h, w) and batch["city"] == ["paris", "pisa"]
    assert D.collate_fn_skip_none([None, None]) is None
    os.remove(samples[1]["img2"])                                               # goes bad after the scan
    assert ds[1] is None
    assert D.collate_fn_skip_none([ds[0], ds[1]])["image1"].shape[0] == 1

    # synthetic layout (what generate_synthetic_data.py:74-81 writes and dataset.py:248-266 scans)
    simg, slab = tmp_path / "synthetic_data" / "images", tmp_path / "synthetic_data" / "labels"
    (simg / "paris").mkdir(parents=True)
    (slab / "paris").mkdir(parents=True)
    for idx in (0, 3):
        Image.fromarray(px1).save(simg / "paris" / f"img1_synth_{idx}.png")
        Image.fromarray(px2).save(simg / "paris" / f"img2_synth_{idx}.png")
        Image.fromarray(lab, mode="L").save(slab / "paris" / f"cm_synth_{idx}.png")
    Image.fromarray(px1).save(simg / "paris" / "img1_synth_7.png")              # img2 missing: skipped
    syn = D.scan_dataset(str(simg), str(slab), is_synthetic=True)
    assert [s["city"] for s in syn] == ["paris_synth", "paris_synth"]
    assert [os.path.basename(s["img2"]) for s in syn] == ["img2_synth_0.png", "img2_synth_3.png"]
    assert [os.path.basename(s["label"]) for s in syn] == ["cm_synth_0.png", "cm_synth_3.png"]
    got = D.BaseChangeDetectionDataset(syn, target_size=(h, w))[1]
    assert torch.equal(got["image1"], want1) and got["city"] == "paris_synth"


def test_launch_list_records_the_request_not_the_answer():
    """_lib.Recorder copies by-reference argument blocks BEFORE the call: pfa_gather_gemm writes the planes it used and the
    number of statistics tiles back into its argument struct, and a replay must pass the original request again."""
    import ctypes
    from pfa_gan import _lib
    _lib.lib()
    rec = _lib.Recorder()
    a = _lib.GatherArgs()
    a.struct_size, a.splitk, a.ntiles_stat = ctypes.sizeof(_lib.GatherArgs), 16, 0
    st, keep = rec.snapshot("pfa_gather_gemm", (ctypes.byref(a), None))
    a.splitk, a.ntiles_stat = 4, 99                       # what the entry point would write back
    copy = [k for k in keep if isinstance(k, _lib.GatherArgs)][0]
    assert (copy.splitk, copy.ntiles_stat) == (16, 0) and st.a0 == ctypes.addressof(copy)
    rec.note_call("pfa_gather_gemm", 0, (st, keep))
    assert len(rec.ops) == 1 and rec.ops[0].kind == _lib.OP_CALL and rec.ops[0].a == ctypes.addressof(st)
