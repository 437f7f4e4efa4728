"""Parity of the HIP ModelTraj path (through the C ABI) with the golden vectors and the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_inf
from trajectory_optimization_amd import synth

pytestmark = pytest.mark.gpu

K, IW, IH = synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT
GRAD_TOL = 1e-5     # north star: gradients within 1e-5 relative (||g-g_ref||_inf / ||g_ref||_inf)
REW_RTOL, REW_ATOL = 2e-5, 2e-6


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _ops():
    from trajectory_optimization_amd import ops
    return ops


def _run_ops(dev, points, poses, quats, rig=None, flags=0, sort=True, clip=(1.0, 5.0)):
    ops = _ops()
    cloud = ops.PackedCloud(torch.from_numpy(np.ascontiguousarray(points)).to(dev), sort=sort)
    cam = ops.Camera(K, IW, IH, clip[0], clip[1])
    p = torch.from_numpy(np.ascontiguousarray(poses)).to(dev)
    q = torch.from_numpy(np.ascontiguousarray(quats)).to(dev)
    rg = ops.CameraRig(rig[0], rig[1], dev) if rig is not None else None
    ws = ops.TrajWorkspace(cloud, p.shape[0] * (rg.n_cams if rg else 1))
    lo_sum, minmax = ops.traj_forward(cloud, p, q, cam, ws, rg, flags=flags)
    rewards, scalars = ops.traj_reward(cloud, lo_sum, cam, ws)
    gout = torch.ones(1, device=dev)
    pg, qg = ops.traj_backward(cloud, p, q, cam, ws, lo_sum, minmax, scalars=scalars, gout=gout, rig=rg, flags=flags)
    torch.cuda.synchronize()
    return dict(lo_sum=lo_sum[:cloud.n].cpu().numpy(), rewards=rewards.cpu().numpy(), minmax=minmax.cpu().numpy(),
                scalars=scalars.cpu().numpy(), pg=pg.cpu().numpy(), qg=qg.cpu().numpy())


def test_wave_reduce_selftest(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(0)
    m = torch.randn(64, 37, generator=g).to(dev)
    s, mn, mx = ops.selftest_wave_reduce(m)
    torch.cuda.synchronize()
    np.testing.assert_allclose(s.cpu().numpy(), m.double().sum(0).cpu().numpy(), rtol=1e-5, atol=1e-5)
    assert torch.equal(mn, m.min(0).values) and torch.equal(mx, m.max(0).values)


def test_packed_cloud_is_a_permutation(dev):
    ops = _ops()
    pts = synth.make_cloud(70_001, seed=2)
    cloud = ops.PackedCloud(torch.from_numpy(pts).to(dev))
    perm = cloud.perm.cpu().numpy()
    n, npad = cloud.n, cloud.npad
    assert np.array_equal(np.sort(perm[:n]), np.arange(n)) and np.all(perm[n:] == -1)
    soa = cloud.soa.cpu().numpy().reshape(3, npad)
    assert np.array_equal(soa[:, :n].T, pts[perm[:n]])
    assert np.array_equal(soa[:, n:], np.repeat(soa[:, n - 1:n], npad - n, axis=1))
    unsorted = ops.PackedCloud(torch.from_numpy(pts).to(dev), sort=False)
    assert np.array_equal(unsorted.perm.cpu().numpy()[:n], np.arange(n))


GOLD = ["traj_bundled_tilted_all", "traj_synth_1000x3", "traj_synth_10000x8", "traj_synth_20000x32",
        "traj_synth_ties", "traj_synth_dense", "traj_synth_clip"]


@pytest.mark.parametrize("name", GOLD)
def test_c_abi_vs_golden(dev, name):
    d = load_golden(name)
    clip = (float(d["min_dist"]), float(d["max_dist"])) if "min_dist" in d else (1.0, 5.0)  # pc_clip_limits of the fixture
    r = _run_ops(dev, d["points"], d["poses"], d["quats"], clip=clip)
    for flags in (0, _ops().DENSE):
        assert np.array_equal(_run_ops(dev, d["points"], d["poses"], d["quats"], clip=clip, flags=flags)["pg"], r["pg"])
    assert abs(r["scalars"][1] - float(d["loss_vis"])) <= 3e-6 * float(d["loss_vis"])
    np.testing.assert_allclose(r["rewards"], d["rewards"], rtol=REW_RTOL, atol=REW_ATOL)
    assert rel_inf(r["pg"], d["vis_poses_grad"]) < GRAD_TOL
    assert rel_inf(r["qg"], d["vis_quats_grad"]) < GRAD_TOL


@pytest.mark.parametrize("name", ["traj_synth_ties", "traj_synth_dense", "traj_synth_20000x32", "traj_bundled_tilted_all"])
def test_culling_is_bitwise_exact(dev, name):
    """The default path skips pairs that provably contribute nothing; TOHIP_TRAJ_DENSE evaluates every pair.
    Same bits out (fixtures incl. ties at the max and a dense cloud with min p > 0)."""
    ops = _ops()
    d = load_golden(name)
    a = _run_ops(dev, d["points"], d["poses"], d["quats"])
    b = _run_ops(dev, d["points"], d["poses"], d["quats"], flags=ops.DENSE)
    for k in ("lo_sum", "rewards", "minmax", "scalars", "pg", "qg"):
        assert np.array_equal(a[k], b[k]), k
    # caller's order is independent of the internal Morton order (sums differ only by association)
    c = _run_ops(dev, d["points"], d["poses"], d["quats"], sort=False)
    np.testing.assert_allclose(c["rewards"], a["rewards"], rtol=1e-6, atol=1e-7)
    assert rel_inf(c["pg"], a["pg"]) < 2e-6 and rel_inf(c["qg"], a["qg"]) < 2e-6


def test_culling_is_bitwise_exact_rig_and_large(dev):
    ops = _ops()
    pts = synth.make_cloud(300_000, seed=51)
    poses, quats = synth.make_path(12, optical=True, jitter_seed=51)
    rq, rt = synth.camera_rig(5)
    for rig in (None, (rq, rt)):
        a = _run_ops(dev, pts, poses, quats, rig=rig)
        b = _run_ops(dev, pts, poses, quats, rig=rig, flags=ops.DENSE)
        for k in ("lo_sum", "rewards", "minmax", "scalars", "pg", "qg"):
            assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("n,w,seed", [(50_000, 16, 31), (200_000, 8, 32), (600_000, 4, 33), (1_000_003, 3, 34)])
def test_c_abi_vs_oracle(dev, n, w, seed):
    """Sizes that exercise every points-per-lane variant (1, 2, 4) and a ragged last tile."""
    from oracle import oracle
    pts = synth.make_cloud(n, seed=seed)
    poses, quats = synth.make_path(w, optical=True, jitter_seed=seed)
    r = _run_ops(dev, pts, poses, quats)
    f = oracle.traj_forward(pts, poses, quats, K, IW, IH, prec="f64")
    pg, qg = oracle.traj_backward(pts, poses, quats, K, IW, IH, f, prec="f64")
    assert abs(r["scalars"][1] - f["loss_vis"]) <= 3e-6 * f["loss_vis"]
    np.testing.assert_allclose(r["rewards"], f["rewards"], rtol=REW_RTOL, atol=REW_ATOL)
    np.testing.assert_allclose(r["minmax"][:, 0], f["pmin"], rtol=1e-5, atol=1e-30)
    np.testing.assert_allclose(r["minmax"][:, 1], f["pmax"], rtol=1e-5)
    assert rel_inf(r["pg"], pg) < GRAD_TOL
    assert rel_inf(r["qg"], qg) < GRAD_TOL


def test_rig_equals_explicit_virtual_waypoints(dev):
    """A C-camera rig == the same cameras written out as W*C independent waypoints (rewards), and the body
    gradient == the chain rule through the composition (checked against finite differences of the oracle)."""
    from oracle import oracle
    pts = synth.make_cloud(30_000, seed=41)
    poses, quats = synth.make_path(4, optical=True, jitter_seed=41)
    rq, rt = synth.camera_rig(3)
    rt = rt + np.array([[0.1, 0.0, 0.2], [0.0, 0.15, 0.2], [-0.1, 0.0, 0.25]], dtype=np.float32)
    r = _run_ops(dev, pts, poses, quats, rig=(rq, rt))

    def virtual(poses, quats):
        qn = quats.astype(np.float64) / np.linalg.norm(quats.astype(np.float64), axis=1, keepdims=True)
        vq = synth.quat_mul(qn[:, None, :], rq[None].astype(np.float64)).reshape(-1, 4)
        R = np.stack([_rot(q) for q in qn])
        vt = (poses.astype(np.float64)[:, None, :] + np.einsum("wij,cj->wci", R, rt.astype(np.float64))).reshape(-1, 3)
        return vt.astype(np.float32), vq.astype(np.float32)

    vt, vq = virtual(poses, quats)
    f = oracle.traj_forward(pts, vt, vq, K, IW, IH, prec="f64")
    np.testing.assert_allclose(r["rewards"], f["rewards"], rtol=5e-5, atol=5e-6)
    assert abs(r["scalars"][1] - f["loss_vis"]) <= 1e-5 * f["loss_vis"]
    # body-pose gradient by central differences of the oracle's loss (coarse: 2e-2 relative)
    h = 1e-3
    for (wi, ci) in [(0, 0), (2, 1), (3, 2)]:
        pp, pm = poses.copy(), poses.copy()
        pp[wi, ci] += h
        pm[wi, ci] -= h
        lp = oracle.traj_forward(pts, *virtual(pp, quats), K, IW, IH, prec="f64")["loss_vis"]
        lm = oracle.traj_forward(pts, *virtual(pm, quats), K, IW, IH, prec="f64")["loss_vis"]
        fd = (lp - lm) / (2 * h)
        assert abs(r["pg"][wi, ci] - fd) <= 0.05 * np.abs(r["pg"]).max() + 1e-7


def _rot(q):
    w, x, y, z = q
    return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def test_full_size_properties(dev):
    """BASELINE.json config 3 (1 M x 128): size-independent properties."""
    ops = _ops()
    n, w = 1_000_000, 128
    pts = synth.make_cloud(n, seed=0)
    poses, quats = synth.make_path(w, optical=True)
    r1 = _run_ops(dev, pts, poses, quats)
    r2 = _run_ops(dev, pts, poses, quats)
    r3 = _run_ops(dev, pts, poses, quats, flags=ops.DENSE)
    # deterministic: fixed-order reductions, bitwise reproducible; exact culling == dense evaluation
    for k in ("rewards", "pg", "qg", "scalars", "minmax"):
        assert np.array_equal(r1[k], r2[k]), k
        assert np.array_equal(r1[k], r3[k]), k
    # every log-odds term is >= 0, so rewards live in [0.5, 1] (f32 sigmoid saturates to exactly 1)
    assert r1["rewards"].min() >= 0.5 and r1["rewards"].max() <= 1.0
    assert 0.5 < r1["rewards"].mean() < 0.9
    assert np.isfinite(r1["pg"]).all() and np.isfinite(r1["qg"]).all() and np.abs(r1["pg"]).max() > 0
    # additivity of the log-odds over waypoint shards (what the multi-GPU all-reduce relies on)
    cloud = ops.PackedCloud(torch.from_numpy(pts).to(dev))
    cam = ops.Camera(K, IW, IH)
    p, q = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
    ws = ops.TrajWorkspace(cloud, 64)
    a, _ = ops.traj_forward(cloud, p[:64].contiguous(), q[:64].contiguous(), cam, ws)
    a = a.clone()
    b, _ = ops.traj_forward(cloud, p[64:].contiguous(), q[64:].contiguous(), cam, ws)
    np.testing.assert_allclose((a + b)[:n].cpu().numpy(), r1["lo_sum"], rtol=1e-5, atol=1e-5)
    # quaternion gradient is tangent to the unit sphere: <q, dL/dq> = 0 (F.normalize)
    dots = (quats.astype(np.float64) * r1["qg"]).sum(1)
    assert np.abs(dots).max() <= 1e-5 * np.abs(r1["qg"]).max()
    # sampled parity against the f64 oracle on a 1/16 subsample of the waypoints at full N
    from oracle import oracle
    sel = np.arange(0, w, 16)
    rs = _run_ops(dev, pts, poses[sel], quats[sel])
    f = oracle.traj_forward(pts, poses[sel], quats[sel], K, IW, IH, prec="f64")
    pg, qg = oracle.traj_backward(pts, poses[sel], quats[sel], K, IW, IH, f, prec="f64")
    np.testing.assert_allclose(rs["rewards"], f["rewards"], rtol=REW_RTOL, atol=REW_ATOL)
    assert rel_inf(rs["pg"], pg) < GRAD_TOL and rel_inf(rs["qg"], qg) < GRAD_TOL


@pytest.mark.parametrize("seed", list(range(12)))
def test_culling_randomized_bitwise(dev, seed):
    """Random clouds (uniform slabs, clustered blobs, thin walls), random paths and cameras: exact culling must
    reproduce the dense evaluation bit for bit (the tile bounds and distance bounds are conservative)."""
    ops = _ops()
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([300, 5_000, 40_000, 150_000, 600_000]))
    kind = seed % 3
    if kind == 0:
        pts = (rng.random((n, 3)) * rng.uniform(5, 60, 3) - rng.uniform(2, 30, 3)).astype(np.float32)
    elif kind == 1:
        centres = rng.uniform(-15, 15, (6, 3))
        pts = (centres[rng.integers(0, 6, n)] + rng.standard_normal((n, 3)) * rng.uniform(0.05, 2.0)).astype(np.float32)
    else:
        pts = np.stack([rng.uniform(-20, 20, n), rng.uniform(-0.02, 0.02, n) + 4.0, rng.uniform(-3, 3, n)], 1).astype(np.float32)
    w = int(rng.integers(3, 40))
    poses = rng.uniform(-8, 8, (w, 3)).astype(np.float32)
    quats = rng.standard_normal((w, 4)).astype(np.float32)
    a = _run_ops(dev, pts, poses, quats)
    b = _run_ops(dev, pts, poses, quats, flags=ops.DENSE)
    for k in ("lo_sum", "rewards", "minmax", "scalars", "pg", "qg"):
        assert np.array_equal(a[k], b[k], equal_nan=True), (k, n, w)


@pytest.mark.parametrize("n,w,cams,occ", [(1_000_000, 24, 1, False), (200_000, 16, 1, True), (60_000, 9, 3, False), (3000, 5, 1, False)])
def test_split_backward_is_bitwise_the_fused_one(dev, n, w, cams, occ):
    """tohip_traj_backward_scan + tohip_traj_backward(need_mask) == tohip_traj_backward, bit for bit (every P, rig,
    occlusion bits), with a general dL/d rewards vector and with the fused visibility loss."""
    ops = _ops()
    pts = synth.make_cloud(n, seed=3)
    poses, quats = synth.make_path(w, optical=True, jitter_seed=2)
    P = torch.from_numpy(pts).to(dev)
    cloud = ops.PackedCloud(P)
    cam = ops.Camera(K, IW, IH)
    p, q = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
    rg = ops.CameraRig(*synth.camera_rig(cams), dev) if cams > 1 else None
    ws = ops.TrajWorkspace(cloud, w * cams)
    bits = ops.occlusion_bits(cloud, P, p, q, cam, 1.0, 15.0, "zbuffer") if occ else None
    lo_sum, minmax = ops.traj_forward(cloud, p, q, cam, ws, rg, flags=ops.DENSE, occ=bits)
    rewards, scalars = ops.traj_reward(cloud, lo_sum, cam, ws)
    gout = torch.ones(1, device=dev)
    g = torch.rand(n, generator=torch.Generator().manual_seed(1)).to(dev) - 0.3
    for kw in (dict(scalars=scalars, gout=gout), dict(grad_rewards=g)):
        ref = ops.traj_backward(cloud, p, q, cam, ws, lo_sum, minmax, rig=rg, flags=ops.DENSE, occ=bits, **kw)
        mask = ops.traj_backward_scan(cloud, p, q, cam, ws, minmax, rig=rg, flags=ops.DENSE, occ=bits)
        assert 0 < int(mask.count_nonzero()) < mask.numel()
        got = ops.traj_backward(cloud, p, q, cam, ws, lo_sum, minmax, rig=rg, flags=ops.DENSE, occ=bits, need_mask=mask, **kw)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    from trajectory_optimization_amd import _lib
    with pytest.raises(_lib.HipError):  # the split belongs to the dense mode
        ops.traj_backward_scan(cloud, p, q, cam, ws, minmax, rig=rg, flags=0, occ=bits)


@pytest.mark.parametrize("n,w,cams,occ", [(1_000_000, 24, 1, False), (200_000, 70, 1, True), (60_000, 9, 3, False), (3000, 5, 1, False),
                                         (300_000, 130, 1, False)])
@pytest.mark.parametrize("dense", [False, True])
def test_backward_with_the_forwards_need_mask_is_bitwise_the_same(dev, n, w, cams, occ, dense):
    """tohip_traj_forward(need_mask_out) -> tohip_traj_backward(need_mask): same gradients, bit for bit, as the backward
    that finds the active pairs itself — both modes, every points-per-lane variant, rig, occlusion bits, more than 64 and
    more than 128 waypoints (mask words), fused loss and a general dL/d rewards vector."""
    ops = _ops()
    pts = synth.make_cloud(n, seed=5)
    poses, quats = synth.make_path(w, optical=True, jitter_seed=4)
    P = torch.from_numpy(pts).to(dev)
    cloud = ops.PackedCloud(P)
    cam = ops.Camera(K, IW, IH)
    p, q = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
    rg = ops.CameraRig(*synth.camera_rig(cams), dev) if cams > 1 else None
    ws = ops.TrajWorkspace(cloud, w * cams)
    bits = ops.occlusion_bits(cloud, P, p, q, cam, 1.0, 15.0, "zbuffer") if occ else None
    flags = ops.DENSE if dense else 0
    lo_ref, mm_ref = ops.traj_forward(cloud, p, q, cam, ws, rg, flags=flags, occ=bits)
    lo_sum, minmax, need = ops.traj_forward(cloud, p, q, cam, ws, rg, flags=flags, occ=bits, want_need=True)
    assert torch.equal(lo_sum, lo_ref) and torch.equal(minmax, mm_ref)  # recording the mask does not change the forward
    if dense:
        scan = ops.traj_backward_scan(cloud, p, q, cam, ws, minmax, rig=rg, flags=flags, occ=bits)
        P_lane = 4 if n >= 512 * 1024 else (2 if n >= 128 * 1024 else 1)  # points per lane the library picks
        used = ((w * cams + 63) // 64) * (cloud.npad // (64 * P_lane)) * 8   # mask words in use (the buffers are padded)
        assert torch.equal(scan[:used], need[:used])  # the same predicate, evaluated by the two kernels
    rewards, scalars = ops.traj_reward(cloud, lo_sum, cam, ws)
    gout = torch.ones(1, device=dev)
    g = torch.rand(n, generator=torch.Generator().manual_seed(2)).to(dev) - 0.3
    for kw in (dict(scalars=scalars, gout=gout), dict(grad_rewards=g)):
        ref = ops.traj_backward(cloud, p, q, cam, ws, lo_sum, minmax, rig=rg, flags=flags, occ=bits, **kw)
        got = ops.traj_backward(cloud, p, q, cam, ws, lo_sum, minmax, rig=rg, flags=flags, occ=bits, need_mask=need, **kw)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])


@pytest.mark.parametrize("seed", range(10))
def test_all_backward_variants_agree_on_random_configs(dev, seed):
    """Random clouds (compact ones with min p > 0 included), paths, rigs, clip limits, unsorted packing, duplicated points:
    culled == dense, and the backward with the forward's record / with the scan's record == the backward without, bit for
    bit.  (tools/stress_bitwise.py runs more of the same.)"""
    ops = _ops()
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([700, 5000, 40_000, 140_000, 300_000]))
    cams = int(rng.choice([1, 1, 2, 5]))
    w = max(1, int(rng.integers(1, 90)) // cams)
    scale = float(rng.choice([0.3, 0.3, 1.0, 2.5]))  # 0.3: everything in front of the cameras -> min p > 0 at some waypoints
    pts = (synth.make_cloud(n, seed=int(rng.integers(1 << 30))) * np.float32(scale)).astype(np.float32)
    if rng.random() < 0.4:
        pts = np.concatenate([pts, pts[: n // 5]])
    poses, quats = synth.make_path(w, optical=True, jitter_seed=int(rng.integers(1 << 30)))
    quats = (quats * np.float32(rng.uniform(0.5, 2.0))).astype(np.float32)
    P = torch.from_numpy(pts).to(dev)
    cloud = ops.PackedCloud(P, sort=bool(rng.random() < 0.8))
    cam = ops.Camera(K, IW, IH, float(rng.uniform(0.2, 2.0)), float(rng.uniform(3.0, 12.0)))
    p, q = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
    rg = ops.CameraRig(*synth.camera_rig(cams), dev) if cams > 1 else None
    ws = ops.TrajWorkspace(cloud, w * cams)
    gout = torch.ones(1, device=dev)

    def same(x, y):
        return torch.equal(torch.isnan(x), torch.isnan(y)) and torch.equal(torch.nan_to_num(x), torch.nan_to_num(y))

    grads = []
    for flags in (0, ops.DENSE):
        lo, mm, need = ops.traj_forward(cloud, p, q, cam, ws, rg, flags=flags, want_need=True)
        rew, sc = ops.traj_reward(cloud, lo, cam, ws)
        plain = ops.traj_backward(cloud, p, q, cam, ws, lo, mm, rig=rg, flags=flags, scalars=sc, gout=gout)
        masked = ops.traj_backward(cloud, p, q, cam, ws, lo, mm, rig=rg, flags=flags, scalars=sc, gout=gout, need_mask=need)
        assert same(plain[0], masked[0]) and same(plain[1], masked[1]), ("forward's record", flags)
        if flags:
            scan = ops.traj_backward_scan(cloud, p, q, cam, ws, mm, rig=rg, flags=flags)
            via_scan = ops.traj_backward(cloud, p, q, cam, ws, lo, mm, rig=rg, flags=flags, scalars=sc, gout=gout, need_mask=scan)
            assert same(plain[0], via_scan[0]) and same(plain[1], via_scan[1]), "scan's record"
        grads.append((lo, mm, rew, plain[0], plain[1]))
    for a, b in zip(grads[0], grads[1]):
        assert same(a, b)  # culled == dense
