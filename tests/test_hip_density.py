"""The dense-scene end of the workload: 1 M points in a small room flag ~6x the (slot, waypoint) pairs of the BASELINE slab, and
the kernels after pass 1 do ~6x the work (bench.py's density_sweep times it).  Parity there: the densest setting of the sweep against
the f64 oracle, the fused step against the split calls, dense == culled bitwise."""
import warnings

import numpy as np
import pytest
import torch

from conftest import rel_inf
from test_hip_conditioning import MARGIN, _margins
from trajectory_optimization_amd import synth

pytestmark = pytest.mark.gpu
K, IW, IH = synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT


@pytest.mark.parametrize("extent,w", [((6.0, 6.0, 3.0), 128), ((10.0, 10.0, 4.0), 48)])
def test_dense_room_against_the_oracle(extent, w):
    from oracle import oracle
    from trajectory_optimization_amd import ops
    dev = torch.device("cuda:0")
    n = 1_000_000
    pts = synth.make_cloud(n, seed=0, extent=extent)
    poses, quats = synth.make_path(w, optical=True, scale=extent[0] / 40.0)
    cloud = ops.PackedCloud(torch.from_numpy(pts).to(dev))
    cam = ops.Camera(K, IW, IH)
    p, q = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
    ws = ops.TrajWorkspace(cloud, w)
    gout = torch.ones(1, device=dev)
    outs = {}
    for name, flags in (("culled", 0), ("dense", ops.DENSE)):
        outs[name] = ops.traj_forward_backward(cloud, p, q, cam, ws, gout, flags=flags)
    st = ops.traj_step_stats(cloud, ws)
    assert st["flagged_fraction"] > 0.03, st   # the setting is what it claims to be: > 4x the BASELINE slab's 0.7 %
    assert all(torch.equal(a, b) for a, b in zip(outs["dense"][:5], outs["culled"][:5]))
    # the split calls (what a sharded run does around its all-reduce) give the same rewards / scalars bit for bit
    lo, _ = ops.traj_forward(cloud, p, q, cam, ws)
    rew, sc, pg2, qg2 = ops.traj_reward_backward(cloud, w, cam, ws, lo, gout)
    rewards, scalars, pg, qg = (t.cpu().numpy() for t in outs["culled"][:4])
    assert np.array_equal(rew.cpu().numpy(), rewards) and np.array_equal(sc.cpu().numpy(), scalars)
    assert rel_inf(pg2.cpu().numpy(), pg) < 1e-6 and rel_inf(qg2.cpu().numpy(), qg) < 1e-6
    f = oracle.traj_forward(pts, poses, quats, K, IW, IH, prec="f64")
    pgo, qgo = oracle.traj_backward(pts, poses, quats, K, IW, IH, f, prec="f64")
    np.testing.assert_allclose(rewards, f["rewards"], rtol=1e-5, atol=0)   # north star: rewards within 1e-5 relative
    assert abs(scalars[1] - f["loss_vis"]) <= 2e-6 * f["loss_vis"]
    # gradients: the 1e-5 bar on every waypoint none of whose points sits within f32 rounding of a threshold of the clipped
    # log-odds (tests/test_hip_conditioning.py): 9 000 points per cubic metre put ~1e5 points per waypoint between p_hat 0.4 and 0.6,
    # so a fair share of the waypoints has one within 3e-7 of 1/2 — one point's on/off is 1e-5 .. 1e-4 of a waypoint's gradient
    keep = _margins(pts, poses, quats, (1.0, 5.0)) > MARGIN
    ep = np.abs(pg - pgo).max(axis=1) / np.abs(pgo).max()
    eq = np.abs(qg - qgo).max(axis=1) / np.abs(qgo).max()
    assert keep.sum() >= w // 2, keep.sum()
    assert (ep[keep] < 1e-5).all() and (eq[keep] < 1e-5).all(), (ep[keep].max(), eq[keep].max())
    # and the others are off by single points, not by anything systematic: by no more than the points within 2 MARGIN of p_hat = 1/2
    # are worth (the f64 oracle with the activity threshold moved down / up by that much; tests/test_hip_reference_dense.py shows on
    # reference-generated fixtures that the reference's own f32 gradient is as undecided on such waypoints)
    if not keep.all():
        lo_g = oracle.traj_backward(pts, poses, quats, K, IW, IH, f, prec="f64", act_shift=-2 * MARGIN)
        hi_g = oracle.traj_backward(pts, poses, quats, K, IW, IH, f, prec="f64", act_shift=2 * MARGIN)
        for g, go, a_lo, a_hi in ((pg, pgo, lo_g[0], hi_g[0]), (qg, qgo, lo_g[1], hi_g[1])):
            band = np.abs(a_lo - a_hi).max(axis=1)
            err = np.abs(g - go).max(axis=1)
            assert (err[~keep] <= 1.05 * band[~keep] + 1e-5 * np.abs(go).max()).all(), (err[~keep], band[~keep])
    warnings.warn(f"dense room {extent}: {int((~keep).sum())} of {w} waypoints have a point within {MARGIN:g} of a threshold and were excluded from "
                  f"the 1e-5 gradient bar (worst of them {max(ep.max(), eq.max()):.1e}); worst among the others {max(ep[keep].max(), eq[keep].max()):.1e}")


def test_culled_pass_evaluates_a_few_percent_of_the_pairs():
    """What the culled pass 1 is for: on the BASELINE slab the probe's lists hold 2-3 % of the (slot, waypoint) pairs, every flagged
    pair among them (tohip_traj_step_stats), and the candidates the sparse kernel walks are the slots that hold one."""
    from trajectory_optimization_amd import ops
    dev = torch.device("cuda:0")
    n, w = 300_000, 64
    cloud = ops.PackedCloud(torch.from_numpy(synth.make_cloud(n, seed=3)).to(dev))
    poses, quats = synth.make_path(w, optical=True)
    cam = ops.Camera(K, IW, IH)
    p, q = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
    ws = ops.TrajWorkspace(cloud, w)
    gout = torch.ones(1, device=dev)
    culled = ops.traj_forward_backward(cloud, p, q, cam, ws, gout, flags=0)
    st = ops.traj_step_stats(cloud, ws)
    dense = ops.traj_forward_backward(cloud, p, q, cam, ws, gout, flags=ops.DENSE)
    sd = ops.traj_step_stats(cloud, ws)
    assert all(torch.equal(a, b) for a, b in zip(dense[:5], culled[:5]))
    total = st["slots"] * st["virtual_waypoints"]
    assert st["flagged_pairs"] == sd["flagged_pairs"] and st["candidate_slots"] == sd["candidate_slots"]
    assert st["flagged_pairs"] <= st["evaluated_pairs"] <= 0.08 * total, st
    assert 0 < st["candidate_slots"] <= st["slots"]


def test_beyond_the_probes_slot_limit_every_pair_is_evaluated():
    """More than 65 536 slots (16.7 M points): a waypoint's row of reachable-slot bits does not fit pass 1's LDS, and the default mode evaluates
    every pair like TOHIP_TRAJ_DENSE does — the same results, no failure."""
    from trajectory_optimization_amd import ops
    dev = torch.device("cuda:0")
    n, w = 16_800_000, 2
    rng = np.random.default_rng(5)
    pts = torch.from_numpy((rng.random((n, 3), dtype=np.float32) * np.float32([60, 60, 4]) - np.float32([30, 30, 2])))
    cloud = ops.PackedCloud(pts.to(dev))
    assert cloud.npad // 256 > 65536
    poses, quats = synth.make_path(w, optical=True)
    cam = ops.Camera(K, IW, IH)
    p, q = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
    ws = ops.TrajWorkspace(cloud, w)
    gout = torch.ones(1, device=dev)
    a = ops.traj_forward_backward(cloud, p, q, cam, ws, gout, flags=0)
    b = ops.traj_forward_backward(cloud, p, q, cam, ws, gout, flags=ops.DENSE)
    assert all(torch.equal(x, y) for x, y in zip(a[:5], b[:5]))
    assert 0.5 < float(a[1][0]) < 0.6 and torch.isfinite(a[2]).all() and float(a[2].abs().max()) > 0


def test_culling_beyond_one_word_block_of_slots():
    """3 M points = 11 719 slots = 184 words of slot bits: the probe's tests beyond its prefetched batch, the culled pass 1's two-level
    search of the popcount prefix and the sparse kernel's walk over several 64-word chunks of candidate bits — against the dense mode,
    bit for bit, and with fewer pairs evaluated."""
    from trajectory_optimization_amd import ops
    dev = torch.device("cuda:0")
    n, w = 3_000_000, 6
    rng = np.random.default_rng(9)
    pts = torch.from_numpy((rng.random((n, 3), dtype=np.float32) * np.float32([50, 50, 4]) - np.float32([25, 25, 2])))
    cloud = ops.PackedCloud(pts.to(dev))
    assert 64 < (cloud.npad // 256 + 63) // 64 <= 1024
    poses, quats = synth.make_path(w, optical=True)
    cam = ops.Camera(K, IW, IH)
    p, q = torch.from_numpy(poses).to(dev), torch.from_numpy(quats).to(dev)
    ws = ops.TrajWorkspace(cloud, w)
    gout = torch.ones(1, device=dev)
    a = ops.traj_forward_backward(cloud, p, q, cam, ws, gout, flags=0)
    st = ops.traj_step_stats(cloud, ws)
    b = ops.traj_forward_backward(cloud, p, q, cam, ws, gout, flags=ops.DENSE)
    assert all(torch.equal(x, y) for x, y in zip(a[:5], b[:5]))
    assert 0 < st["flagged_pairs"] <= st["evaluated_pairs"] < 0.1 * st["slots"] * st["virtual_waypoints"], st
    # and as two trajectories in one pass (candidate bits of two rows of 184 words)
    toff = torch.tensor([0, 2, w], dtype=torch.int32, device=dev)
    wsm = ops.TrajWorkspace(cloud, w, 2)
    m = ops.traj_forward_backward_multi(cloud, p, q, toff, cam, wsm, torch.ones(2, device=dev), flags=0)
    one0 = ops.traj_forward_backward(cloud, p[:2].contiguous(), q[:2].contiguous(), cam, ops.TrajWorkspace(cloud, 2), gout, flags=0)
    one1 = ops.traj_forward_backward(cloud, p[2:].contiguous(), q[2:].contiguous(), cam, ops.TrajWorkspace(cloud, w - 2), gout, flags=0)
    assert torch.equal(m[0][0], one0[0]) and torch.equal(m[0][1], one1[0])          # rewards per trajectory
    assert torch.equal(m[2][:2], one0[2]) and torch.equal(m[2][2:], one1[2])          # position gradients
    assert torch.equal(m[3][:2], one0[3]) and torch.equal(m[3][2:], one1[3])
