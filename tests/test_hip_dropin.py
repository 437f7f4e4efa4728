"""The reference's own loop — zero_grad(); loss = model(); loss.backward(); optimizer.step()
(/root/reference/src/trajectory_optimization.py:109-116) — over the drop-in classes: the short cuts it takes on this chip (one
library call per direction, backward on the calling thread, one-launch Adam inside torch.optim.Adam.step, a captured step) give
what the plain route gives."""
import numpy as np
import pytest
import torch

from conftest import REPO, load_golden
from trajectory_optimization_amd import synth

pytestmark = pytest.mark.gpu
K, IW, IH = synth.K_INTRINS, synth.IMG_WIDTH, synth.IMG_HEIGHT


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _model(dev, n=60_000, w=12, seed=4, **kw):
    from trajectory_optimization_amd.model import ModelTraj
    pts = synth.make_cloud(n, seed=seed)
    poses, quats = synth.make_path(w, optical=True, jitter_seed=seed)
    return ModelTraj(torch.from_numpy(pts), torch.from_numpy(poses), torch.from_numpy(quats), torch.from_numpy(K), IW, IH, device=dev, **kw)


def _groups(m):
    return [{"params": [m.poses], "lr": 0.1}, {"params": [m.quats], "lr": 0.02}]


@pytest.mark.parametrize("vwd", [0.0, 0.5])
def test_backward_on_the_calling_thread_equals_the_engine(dev, vwd):
    """loss.backward() of the returned loss (no engine) == the same through torch's autograd engine, to the bit; gradients
    accumulate into an existing .grad; a second backward raises like torch's."""
    out = []
    for fast in (True, False):
        m = _model(dev)
        m.fast_backward = fast
        loss = m(vis_wps_dist=vwd)
        assert (type(loss) is not torch.Tensor) == fast
        loss.backward()
        g1 = (m.poses.grad.clone(), m.quats.grad.clone())
        loss2 = m(vis_wps_dist=vwd)
        loss2.backward()   # accumulates
        out.append((loss.detach().clone(), g1, (m.poses.grad.clone(), m.quats.grad.clone())))
        with pytest.raises(RuntimeError, match="second time"):
            loss2.backward()
    assert torch.equal(out[0][0], out[1][0])
    for k in (1, 2):
        assert torch.equal(out[0][k][0], out[1][k][0]) and torch.equal(out[0][k][1], out[1][k][1])
    assert torch.equal(out[0][2][0], 2 * out[0][1][0])


def test_fast_backward_steps_aside(dev):
    """Anything but the plain call goes through torch's engine: an explicit gradient, a hook on a Parameter, a loss built on
    the returned one."""
    m = _model(dev)
    ref = m(vis_wps_dist=0.0)
    ref.backward()
    g0 = m.poses.grad.clone()
    m.zero_grad()
    m(vis_wps_dist=0.0).backward(gradient=torch.tensor(3.0, device=dev))
    np.testing.assert_allclose(m.poses.grad.cpu().numpy(), 3 * g0.cpu().numpy(), rtol=1e-5, atol=1e-8)
    m.zero_grad()
    (2 * m(vis_wps_dist=0.0)).backward()
    np.testing.assert_allclose(m.poses.grad.cpu().numpy(), 2 * g0.cpu().numpy(), rtol=1e-5, atol=1e-8)
    m.zero_grad()
    seen = []
    h = m.poses.register_hook(lambda g: seen.append(g.clone()))
    m(vis_wps_dist=0.0).backward()
    h.remove()
    assert len(seen) == 1 and torch.equal(seen[0], g0)


def test_backward_after_another_forward_rebuilds_its_step(dev):
    """loss_a = model(); loss_b = model(); loss_a.backward(): the second forward reused the workspace; the first step's state is
    rebuilt from its (unchanged) inputs — same gradients as without the second forward."""
    m = _model(dev)
    la = m(vis_wps_dist=0.0)
    la.backward()
    g = (m.poses.grad.clone(), m.quats.grad.clone())
    m.zero_grad()
    la = m(vis_wps_dist=0.0)
    m(vis_wps_dist=0.0)
    la.backward()
    assert torch.equal(m.poses.grad, g[0]) and torch.equal(m.quats.grad, g[1])


def test_general_backward_then_fused_backward_of_the_same_step(dev):
    """model.loss['vis'].backward(retain_graph=True) goes through the general kernels and leaves ITS pair sums in the plan's
    workspace; the loss.backward() that follows must not read them as the unit-gradient sums (tohip_traj_loss_refresh): the
    accumulated gradient equals the one of the op-by-op route."""
    grads = []
    for fused in (True, False):
        m = _model(dev)
        m.fused_loss = fused
        loss = m(vis_wps_dist=0.0)
        m.loss["vis"].backward(retain_graph=True)
        g_vis = (m.poses.grad.clone(), m.quats.grad.clone())
        loss.backward()
        grads.append((g_vis, (m.poses.grad.clone(), m.quats.grad.clone())))
    for k in (0, 1):
        for j in (0, 1):
            a, b = grads[0][k][j].cpu().numpy(), grads[1][k][j].cpu().numpy()
            assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max(), (k, j)
    # and the fused backward alone, after the general one, equals a fresh step's to the bit
    m = _model(dev)
    m(vis_wps_dist=0.0).backward()
    g = (m.poses.grad.clone(), m.quats.grad.clone())
    m.zero_grad()
    loss = m(vis_wps_dist=0.0)
    m.rewards.sum().backward(retain_graph=True)
    m.zero_grad()
    loss.backward()
    assert torch.equal(m.poses.grad, g[0]) and torch.equal(m.quats.grad, g[1])


def test_backward_after_a_step_and_another_forward_raises(dev):
    """la = model(); opt.step(); model(); la.backward(): the one-launch Adam writes the Parameters through raw pointers but bumps
    their version counters, so the rebuild of la's step refuses (its inputs are gone) instead of returning another step's gradient;
    ModelPose's saved (trans, quat) make torch's own in-place check fire."""
    from trajectory_optimization_amd.model import ModelPose
    m = _model(dev)
    opt = torch.optim.Adam(_groups(m))
    m(vis_wps_dist=0.0).backward()
    v0 = m.poses._version
    opt.step()
    assert m.poses._version > v0
    la = m(vis_wps_dist=0.0)
    la.backward(retain_graph=True)
    opt.step()
    m(vis_wps_dist=0.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        la.backward()
    pts = synth.make_cloud(20_000, seed=9)
    mp = ModelPose(torch.from_numpy(pts), torch.tensor([[0.0, 0.0, 0.0]]), torch.tensor([[1.0, 0.0, 0.0, 0.0]]), torch.from_numpy(K), IW, IH, device=dev)
    optp = torch.optim.Adam([{"params": [mp.trans], "lr": 0.1}, {"params": [mp.quat], "lr": 0.1}])
    lp = mp()
    lp.backward(retain_graph=True)
    optp.step()
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        lp.backward()


def test_torch_adam_with_the_one_launch_update(dev):
    """torch.optim.Adam over the models' Parameters: with the step pre-hook (one launch, torch's own state entries) — asked for per
    optimizer INSTANCE or process-wide — and without it the trajectories agree to rounding; gradients are back in .grad after
    step(); the state keeps torch's layout; a configuration the kernel does not cover (weight decay) is left to torch."""
    from trajectory_optimization_amd import optimizer as O
    runs = []
    for accel in ("instance", True, False):
        O.accelerate_torch_adam(accel is True)
        try:
            m = _model(dev)
            opt = torch.optim.Adam(_groups(m))
            if accel == "instance":
                assert O.accelerate_torch_adam(opt) is opt and O.accelerate_torch_adam(opt) is opt   # (asking twice registers once)
                assert len(opt._optimizer_step_pre_hooks) == 1 and O.torch_adam_accelerated()[0] is False
            sched = torch.optim.lr_scheduler.ExponentialLR(optimizer=opt, gamma=0.9)
            for i in range(8):
                opt.zero_grad()
                loss = m(vis_wps_dist=0.0)
                loss.backward()
                g = m.poses.grad
                opt.step()
                assert m.poses.grad is g
                if i == 1:
                    early = m.poses.detach().clone()
                if i % 3 == 0:
                    sched.step()
            st = opt.state[m.poses]
            assert set(st.keys()) == {"step", "exp_avg", "exp_avg_sq"} and float(st["step"]) == 8.0 and not st["step"].is_cuda
            assert ("_tohip_adam_arr" in opt.__dict__) == (accel is not False)      # the one-launch update ran / did not run
            runs.append((m.poses.detach().clone(), m.quats.detach().clone(), early))
        finally:
            O.accelerate_torch_adam(False)
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])   # per instance == process-wide, to the bit
    # Adam's first steps are sign-like (|update| = lr wherever g != 0): 1e-6 differences in g near 0 can move a coordinate by a
    # visible fraction of lr; elsewhere the two agree to float rounding
    np.testing.assert_allclose(runs[0][0].cpu().numpy(), runs[2][0].cpu().numpy(), rtol=0, atol=2e-3)
    np.testing.assert_allclose(runs[0][1].cpu().numpy(), runs[2][1].cpu().numpy(), rtol=0, atol=2e-3)
    assert np.median(np.abs(runs[0][2].cpu().numpy() - runs[2][2].cpu().numpy())) < 1e-6   # after two steps: rounding only
    # switching between the two in mid-run continues on the same state; weight decay is torch's business
    m = _model(dev)
    opt = torch.optim.Adam(_groups(m))
    for accel in (True, False, True):
        O.accelerate_torch_adam(accel)
        opt.zero_grad(); m(vis_wps_dist=0.0).backward(); opt.step()
    assert float(opt.state[m.poses]["step"]) == 3.0
    # an instance with its own hooks under the process-wide switch: updated ONCE per step
    m2 = _model(dev)
    opt2 = O.accelerate_torch_adam(torch.optim.Adam(_groups(m2)))
    opt2.zero_grad(); m2(vis_wps_dist=0.0).backward(); opt2.step()
    assert float(opt2.state[m2.poses]["step"]) == 1.0
    O.accelerate_torch_adam(False)
    m = _model(dev)
    opt = O.accelerate_torch_adam(torch.optim.Adam(_groups(m), weight_decay=0.1))
    p0 = m.poses.detach().clone()
    opt.zero_grad(); m(vis_wps_dist=0.0).backward(); opt.step()
    assert not torch.equal(m.poses.detach(), p0) and "_tohip_adam_arr" not in opt.__dict__


def test_building_a_model_changes_no_global_torch_state():
    """A fresh process: building ModelTraj / ModelPose and running the reference's loop registers NO optimizer hook anywhere (until
    r05 the first model installed two process-wide ones); ModelTraj(..., fast_adam=True) is the spelled-out opt-in."""
    import subprocess
    import sys
    code = """
import sys, torch
sys.path.insert(0, %r)
from torch.optim import optimizer as TO
from trajectory_optimization_amd import synth, optimizer as O
from trajectory_optimization_amd.model import ModelTraj, ModelPose
dev = torch.device('cuda:0')
pts = torch.from_numpy(synth.make_cloud(5000, seed=1))
poses, quats = (torch.from_numpy(a) for a in synth.make_path(5, optical=True))
K = torch.from_numpy(synth.K_INTRINS)
m = ModelTraj(pts, poses, quats, K, synth.IMG_WIDTH, synth.IMG_HEIGHT, device=dev)
mp = ModelPose(pts, torch.zeros(1, 3), torch.tensor([[1.0, 0, 0, 0]]), K, synth.IMG_WIDTH, synth.IMG_HEIGHT, device=dev)
opt = torch.optim.Adam([{'params': [m.poses], 'lr': 0.1}, {'params': [m.quats], 'lr': 0.02}])
opt.zero_grad(); m(vis_wps_dist=0.0).backward(); opt.step()
assert len(TO._global_optimizer_pre_hooks) == 0 and len(TO._global_optimizer_post_hooks) == 0
assert len(opt._optimizer_step_pre_hooks) == 0 and O.torch_adam_accelerated() == (False, False) and '_tohip_adam_arr' not in opt.__dict__
m2 = ModelTraj(pts, poses, quats, K, synth.IMG_WIDTH, synth.IMG_HEIGHT, device=dev, fast_adam=True)
assert len(TO._global_optimizer_pre_hooks) == 1 and O.torch_adam_accelerated() == (True, True)
opt2 = torch.optim.Adam([{'params': [m2.poses], 'lr': 0.1}, {'params': [m2.quats], 'lr': 0.02}])
opt2.zero_grad(); m2(vis_wps_dist=0.0).backward(); opt2.step()
assert '_tohip_adam_arr' in opt2.__dict__
print('ok')
""" % REPO
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-1000:] + r.stderr[-3000:]


def test_adam_trajectory_fixture_through_the_shortcuts(dev):
    """The reference's Adam trajectory (fixture generated from the reference itself) with every short cut on."""
    d = load_golden("traj_adam_bundled")
    from trajectory_optimization_amd.model import ModelTraj
    m = ModelTraj(points=torch.from_numpy(load_golden("bundled")["pts"]), wps_poses=torch.from_numpy(d["poses"]),
                  wps_quats=torch.from_numpy(d["quats"]), intrins=torch.from_numpy(K), img_width=IW, img_height=IH, device=dev)
    from trajectory_optimization_amd.optimizer import accelerate_torch_adam
    opt = accelerate_torch_adam(torch.optim.Adam([{"params": [m.poses], "lr": float(d["lr_pose"])}, {"params": [m.quats], "lr": float(d["lr_quat"])}]))
    for i in range(10):
        opt.zero_grad()
        m().backward()
        opt.step()
        if i + 1 in (1, 5, 10):
            np.testing.assert_allclose(m.poses.detach().cpu().numpy(), d[f"poses_step{i + 1}"], rtol=0, atol=2e-3)
    assert "_tohip_adam_arr" in opt.__dict__


def test_captured_step_equals_the_eager_loop(dev):
    """The whole step — model(), loss.backward(), torch.optim.Adam(capturable=True).step() — captured into a HIP graph
    (torch.cuda.graph) and replayed gives, bit for bit, what the same loop gives eagerly."""
    def loop_eager(m, opt, n):
        for _ in range(n):
            opt.zero_grad()
            loss = m(vis_wps_dist=0.0)
            loss.backward()
            opt.step()
        return loss

    me = _model(dev, n=80_000, w=16)
    oe = torch.optim.Adam(_groups(me), capturable=True)
    loop_eager(me, oe, 3 + 4)
    mg = _model(dev, n=80_000, w=16)
    og = torch.optim.Adam(_groups(mg), capturable=True)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        loop_eager(mg, og, 3)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph()
    og.zero_grad(set_to_none=True)
    with torch.cuda.graph(graph):
        loss = mg(vis_wps_dist=0.0)
        loss.backward()
        og.step()
    # the capture itself does not execute: four replays = steps 4..7
    for _ in range(4):
        graph.replay()
    torch.cuda.synchronize(dev)
    assert torch.equal(mg.poses.detach(), me.poses.detach()) and torch.equal(mg.quats.detach(), me.quats.detach())
    assert torch.equal(mg.rewards.detach(), me.rewards.detach())
    assert torch.equal(mg.poses.grad, me.poses.grad)


@pytest.mark.parametrize("hpr", [False, True])
def test_pose_backward_on_the_calling_thread_equals_the_engine(dev, hpr):
    """ModelPose (/root/reference/src/pose_optimization.py:130-136): loss.backward() of the returned loss without the engine ==
    through the engine, bit for bit; an in-place edit of the pose before backward() raises like torch's saved tensors."""
    from trajectory_optimization_amd.model import ModelPose
    pts = torch.from_numpy(synth.make_cloud(30_000, seed=8))
    out = []
    for fast in (True, False):
        m = ModelPose(pts, torch.tensor([[1.0, 0.5, 0.0]]), torch.tensor([[0.9, 0.1, -0.3, 0.2]]), torch.from_numpy(K), IW, IH, device=dev)
        m.fast_backward = fast
        loss = m(hpr=hpr)
        assert (type(loss) is not torch.Tensor) == fast
        loss.backward()
        m(hpr=hpr).backward()   # accumulates
        out.append((loss.detach().clone(), m.trans.grad.clone(), m.quat.grad.clone(), m.observations.detach().clone()))
        loss = m(hpr=hpr)
        with torch.no_grad():
            m.trans.add_(0.01)
        with pytest.raises(RuntimeError, match="inplace|in-place"):
            loss.backward()
    assert all(torch.equal(a, b) for a, b in zip(*out))
