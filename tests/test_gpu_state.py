"""Device-state bookkeeping of the OTTER object on the GPU: when packed weights are rebuilt, what happens to a running
episode when they are, the reference's plain-attribute cache semantics, integer image dtypes, and captured encoder graphs
replayed at the same time on two streams."""
import numpy as np
import pytest
import torch

import golden_inputs as GI
from oracle import otter_ref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from otter_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _model(mc, sc, seed, dev):
    from otter_amd.otter import OTTER
    from otter_amd.params import otter_specs, random_state_dict
    sd = random_state_dict(otter_specs(mc, sc), seed=seed)
    model = OTTER(mc, sc)
    model.load_state_dict(sd, strict=True)
    return model.to(dev).eval(), sd


def _obs(case, dev, step=0):
    r_images, r_text, r_proprio = GI.rollout_inputs(case)
    return r_images[step].to(dev), r_text, r_proprio[step].to(dev)


def test_weight_writes_reach_the_device_copies(dev):
    case = GI.E2E_CASES[0]
    mc, sc = GI.e2e_configs(case)
    model, sd = _model(mc, sc, case["seed"], dev)
    img, text, prop = _obs(case, dev)

    def first_action():
        model.reset_cache()
        return model.forward_inference(img, text, prop).cpu()

    base = first_action()
    bias = model.action_head.mlp[2].bias
    # (1) a write through the tensor itself bumps its version counter: seen automatically
    with torch.no_grad():
        bias.add_(0.25)
    torch.testing.assert_close(first_action(), base + 0.25, rtol=0, atol=1e-5)
    # (2) re-pointing .data (the reference does this to its CLIP parameters, otter/policy/otter.py:80): seen automatically
    bias.data = bias.data + 0.25
    torch.testing.assert_close(first_action(), base + 0.5, rtol=0, atol=1e-5)
    # (3) an in-place write THROUGH .data changes neither version nor address.  (Tensors the device model uses in place --
    #     fp32 biases and LayerNorm vectors -- show it anyway; a packed matrix does not: the explicit call is needed.)
    weight = model.action_head.mlp[2].weight
    before = first_action()
    weight.data.mul_(1.5)
    assert torch.equal(first_action(), before)                  # stale, as documented
    model.invalidate_device_weights()
    after = first_action()
    assert (after - before).abs().max().item() > 1e-3
    # (4) ... or the opt-in value checksum
    model.check_weights_by_value = True
    first_action()
    weight.data.mul_(1.0 / 1.5)
    torch.testing.assert_close(first_action(), before, rtol=0, atol=1e-5)
    model.check_weights_by_value = False
    dims = R.Dims(first_k_tokens=mc.first_k_tokens, pooling_heads=mc.pooling_heads, transformer_heads=mc.transformer_heads,
                  action_dim=mc.action_dim, action_horizon=sc.action_horizon, seq_length=sc.seq_length)
    now = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ref = R.Rollout(now, dims).step(R.clip_transform(img.cpu()), text, prop.cpu())
    assert (first_action() - ref).abs().max().item() <= 1e-3
    # (5) buffers that the device model packs are part of the key too
    with torch.no_grad():
        model.pe.pe.add_(0.1)
    assert (first_action() - before).abs().max().item() > 1e-4


def test_history_survives_a_weight_change_and_cache_attribute_assignment(dev):
    case = GI.E2E_CASES[0]
    mc, sc = GI.e2e_configs(case)
    model, sd = _model(mc, sc, case["seed"], dev)
    r_images, r_text, r_proprio = GI.rollout_inputs(case)
    model.reset_cache()
    for t in range(2):
        model.forward_inference(r_images[t].to(dev), r_text, r_proprio[t].to(dev))
    before = model.cached_transformer_input.clone()
    assert before.shape[:2] == (1, 2)
    with torch.no_grad():
        model.action_head.mlp[2].bias.add_(1.0)              # e.g. an optimizer step between two calls of one episode
    a2 = model.forward_inference(r_images[2].to(dev), r_text, r_proprio[2].to(dev)).cpu()
    assert model.cache_size == 3                                # the episode went on: history was carried over
    torch.testing.assert_close(model.cached_transformer_input[:, :2], before, rtol=0, atol=0)
    # the same three steps on the oracle with the changed head
    sd2 = dict(sd)
    sd2["action_head.mlp.2.bias"] = sd["action_head.mlp.2.bias"] + 1.0
    dims = R.Dims(first_k_tokens=mc.first_k_tokens, pooling_heads=mc.pooling_heads, transformer_heads=mc.transformer_heads,
                  action_dim=mc.action_dim, action_horizon=sc.action_horizon, seq_length=sc.seq_length)
    ro = R.Rollout(sd2, dims)
    for t in range(3):
        ref = ro.step(R.clip_transform(r_images[t]), r_text, r_proprio[t])
    assert (a2 - ref).abs().max().item() <= 1e-3

    # the reference's cache is a plain attribute (otter/policy/otter.py:149-151): assignment works
    window = model.cached_transformer_input.clone()
    model.cached_transformer_input = None
    assert model.cache_size == 0 and model.cached_transformer_input is None
    model.cached_transformer_input = window
    assert model.cache_size == 3
    torch.testing.assert_close(model.cached_transformer_input, window, rtol=0, atol=0)
    a3 = model.forward_inference(r_images[3].to(dev), r_text, r_proprio[3].to(dev)).cpu()
    ref3 = ro.step(R.clip_transform(r_images[3]), r_text, r_proprio[3])
    assert (a3 - ref3).abs().max().item() <= 1e-3
    with pytest.raises(ValueError):
        model.cached_transformer_input = window[0]


def test_two_invalidations_in_a_row_keep_the_running_episode(dev):
    """Two `invalidate_device_weights()` with no inference call between them (two optimizer steps during an episode): the
    window carried at the first one survives the second.  And a change of a packed buffer reaches the training-loss path
    as it reaches inference."""
    case = GI.E2E_CASES[0]
    mc, sc = GI.e2e_configs(case)
    model, sd = _model(mc, sc, case["seed"], dev)
    r_images, r_text, r_proprio = GI.rollout_inputs(case)
    model.reset_cache()
    for t in range(2):
        model.forward_inference(r_images[t].to(dev), r_text, r_proprio[t].to(dev))
    before = model.cached_transformer_input.clone()
    model.invalidate_device_weights(towers=False)
    model.invalidate_device_weights(towers=False)
    assert model.cached_transformer_input is not None
    torch.testing.assert_close(model.cached_transformer_input, before, rtol=0, atol=0)
    a2 = model.forward_inference(r_images[2].to(dev), r_text, r_proprio[2].to(dev)).cpu()
    assert model.cache_size == 3
    dims = R.Dims(first_k_tokens=mc.first_k_tokens, pooling_heads=mc.pooling_heads, transformer_heads=mc.transformer_heads,
                  action_dim=mc.action_dim, action_horizon=sc.action_horizon, seq_length=sc.seq_length)
    ro = R.Rollout(sd, dims)
    for t in range(3):
        ref = ro.step(R.clip_transform(r_images[t]), r_text, r_proprio[t])
    assert (a2 - ref).abs().max().item() <= 1e-3
    model.invalidate_device_weights(keep_history=False)
    assert model.cached_transformer_input is None
    # the positional table is a buffer both the inference model and the training tail pack: a change reaches both
    images, text, proprio, gt = GI.e2e_inputs(case)
    with torch.no_grad():
        l0 = model(images.to(dev), text, proprio.to(dev), gt.to(dev)).item()
        model.pe.pe.mul_(0.0)
        l1 = model(images.to(dev), text, proprio.to(dev), gt.to(dev)).item()          # the loss path repacks its copy of the table
        sd2 = dict(sd)
        sd2["pe.pe"] = torch.zeros_like(sd["pe.pe"])
        dims_t = R.Dims(first_k_tokens=mc.first_k_tokens, pooling_heads=mc.pooling_heads, transformer_heads=mc.transformer_heads,
                        action_dim=mc.action_dim, action_horizon=sc.action_horizon, seq_length=sc.seq_length, num_cameras=len(sc.camera_keys),
                        pool_true_text=mc.pool_true_text, use_mse=sc.mse_loss)
        ref1 = R.forward(images, text, proprio, sd2, dims_t, gt).item()
    assert abs(l1 - ref1) <= 1e-3 and abs(l0 - l1) > 1e-4


@pytest.mark.parametrize("dtype", [torch.int16, torch.int32, torch.int64])
def test_every_integer_image_dtype_is_normalised(dev, dtype):
    """otter/policy/otter.py:367-368 normalises every non-float image tensor, not only uint8."""
    case = GI.E2E_CASES[0]
    mc, sc = GI.e2e_configs(case)
    model, sd = _model(mc, sc, case["seed"], dev)
    images, text, proprio, gt = GI.e2e_inputs(case)
    assert images.dtype == torch.uint8
    want = model(images.to(dev), text, proprio.to(dev)).cpu()
    got = model(images.to(dtype).to(dev), text, proprio.to(dev)).cpu()
    assert torch.equal(got, want)
    # values outside a byte cannot ride the uint8 route: the elementwise transform itself, against the oracle's
    big = images.to(dtype) * 2
    got = model(big.to(dev), text, proprio.to(dev)).cpu()
    dims = R.Dims(first_k_tokens=mc.first_k_tokens, pooling_heads=mc.pooling_heads, transformer_heads=mc.transformer_heads,
                  action_dim=mc.action_dim, action_horizon=sc.action_horizon, seq_length=sc.seq_length)
    with torch.no_grad():
        ref = R.forward(R.clip_transform(big), text, proprio, sd, dims)
    assert (got - ref).abs().max().item() <= 1e-3


def test_two_encoder_graphs_replayed_on_two_streams(dev):
    """Each captured encoder graph owns its split-K scratch: two rollouts stepping at the same time on different streams
    give what they give one after the other."""
    from otter_amd import engine
    case = GI.E2E_CASES[0]
    mc, sc = GI.e2e_configs(case)
    model, _ = _model(mc, sc, case["seed"], dev)
    dm = model.device_model()
    cams, res = len(model.camera_keys), sc.image_size
    g = torch.Generator().manual_seed(3)
    steps = 12
    inputs = [[(torch.randint(0, 256, (cams, 3, res, res), dtype=torch.uint8, generator=g).to(dev),
                torch.randn(1, mc.first_k_tokens, dm.clip.embed, generator=g).to(dev),
                torch.randn(1, mc.proprio_input_dim, generator=g).to(dev)) for _ in range(steps)] for _ in range(2)]
    serial = []
    for e in range(2):
        ro = engine.BatchedRollout(dm, 1)
        serial.append([ro.step(*inputs[e][t]).clone() for t in range(steps)])
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    ros = [engine.BatchedRollout(dm, 1) for _ in range(2)]
    outs = [[], []]
    for t in range(steps):
        for e in range(2):
            with torch.cuda.stream(streams[e]):
                outs[e].append(ros[e].step(*inputs[e][t]).clone())
    torch.cuda.synchronize()
    assert ros[0]._graph is not None and ros[0]._graph.ws.data_ptr() != ros[1]._graph.ws.data_ptr()
    for e in range(2):
        for t in range(steps):
            assert torch.equal(outs[e][t], serial[e][t]), (e, t)


@pytest.mark.parametrize("receding", [False, True])
def test_device_action_queue_matches_the_host_queue(dev, receding):
    """convert_abs_action + temporal ensemble / receding horizon + 10-d -> 7-d for 9 episodes in one launch per control
    step, against the single-episode host code (numpy float64, itself held to the reference's outputs) run per episode;
    a subset of the episodes restarts in the middle."""
    from otter_amd.action_utils import convert_abs_action
    from otter_amd.rollout_server import ActionQueue, DeviceActionQueue, poses_to_proprio, to_robot_action
    n, H, X, steps = 9, 12, 5, 17
    rng = np.random.default_rng(4)
    dq = DeviceActionQueue(n, X, dev, receding_horizon_control=receding)
    hq = [ActionQueue(X, receding) for _ in range(n)]
    worst7 = worst10 = worst_abs = 0.0
    for s in range(steps):
        if s == 8:
            dq.clear([2, 5])
            hq[2].clear(); hq[5].clear()
        delta = rng.standard_normal((n, H, 10)).astype(np.float32) * 0.3
        delta[..., 3:9] += np.array([1, 0, 0, 0, 1, 0], dtype=np.float32)          # rotations near the identity, not orthonormal
        delta[..., 9] = rng.uniform(0, 1, (n, H))
        pose = poses_to_proprio(rng.standard_normal((n, 6)) * 0.5, rng.uniform(-0.2, 1.2, n))
        robot, act10, abs_chunks = dq.step(torch.from_numpy(delta).to(dev), torch.from_numpy(pose).to(dev), want_abs=True)
        robot, act10, abs_chunks = robot.cpu().numpy(), act10.cpu().numpy(), abs_chunks.cpu().numpy()
        for e in range(n):
            # (float64 in: on float32 chunks the host orthonormalises the delta rotation in float32, 2e-7 away from this)
            ref_abs = convert_abs_action(delta[e][None].astype(np.float64), pose[e][None, None])[0]      # what OtterInference returns
            worst_abs = max(worst_abs, float(np.abs(abs_chunks[e] - ref_abs).max()))
            queue = hq[e]
            ref10 = queue.next_action(ref_abs) if (not receding or queue.needs_prediction()) else queue.next_action(None)
            worst10 = max(worst10, float(np.abs(act10[e] - ref10).max()))
            ref7 = to_robot_action(ref10)                      # (orthonormalises the rotation rows of its argument in place, as upstream)
            worst7 = max(worst7, float(np.abs(robot[e] - ref7).max()))
            assert robot[e, 6] == ref7[6] and robot[e, 6] in (0.0, 1.0)
    print(f"device action queue (receding={receding}): abs chunks {worst_abs:.1e}, executed 10-d {worst10:.1e}, robot 7-d {worst7:.1e}")
    assert worst_abs <= 1e-12 and worst10 <= 1e-12 and worst7 <= 1e-9
    assert dq.counts.cpu().tolist() == [steps if e not in (2, 5) else steps - 8 for e in range(n)]


def test_batched_rollout_policy_end_to_end(dev):
    """Frames + poses of 3 concurrent episodes -> 7-d robot actions, everything between on the device; episode 1 against
    the single-episode chain OtterInference-style decode + host ActionQueue fed from the same model."""
    from otter_amd.action_utils import convert_abs_action
    from otter_amd.rollout_server import ActionQueue, BatchedRolloutPolicy, poses_to_proprio, to_robot_action
    case = GI.E2E_CASES[0]
    mc, sc = GI.e2e_configs(case)
    model, sd = _model(mc, sc, case["seed"], dev)
    n, X = 3, 3
    pol = BatchedRolloutPolicy(model, n, X)
    rng = np.random.default_rng(12)
    text = GI._tokens(rng, n, (4, 6, 9))
    queue = ActionQueue(X)
    model.reset_cache()
    for s in range(sc.seq_length + 2):
        frames = torch.from_numpy(rng.integers(0, 256, (n * 2, 3, sc.image_size, sc.image_size), dtype=np.uint8)).to(dev)
        cart, grip = rng.standard_normal((n, 6)) * 0.4, rng.uniform(0, 1, n)
        got = pol(frames, text, cart, grip).cpu().numpy()
        pose = poses_to_proprio(cart, grip)
        chunk = model.forward_inference(frames[2:4], text[1], torch.from_numpy(pose[1]).float().to(dev)).cpu().numpy()
        ref = to_robot_action(queue.next_action(convert_abs_action(chunk[None].astype(np.float64), pose[1][None, None])[0]))
        assert got.shape == (n, 7) and np.abs(got[1] - ref).max() <= 1e-4
    pol.start_episodes([1])
    assert pol.queue.counts.cpu().tolist() == [sc.seq_length + 2, 0, sc.seq_length + 2] and pol.rollout.lens_host[1] == 0


def test_batched_rollout_policy_follows_weight_updates(dev):
    """A weight written between two control steps of running episodes reaches the batched policy on its next call (as it
    reaches OTTER.forward_inference), and the episodes' histories move over to the new weights."""
    from otter_amd.rollout_server import BatchedRolloutPolicy
    case = GI.E2E_CASES[0]
    mc, sc = GI.e2e_configs(case)
    model, sd = _model(mc, sc, case["seed"], dev)
    n = 2
    pol = BatchedRolloutPolicy(model, n, 2)
    rng = np.random.default_rng(14)
    text = GI._tokens(rng, n, (4, 6))
    model.reset_cache()

    def call():
        frames = torch.from_numpy(rng.integers(0, 256, (n * 2, 3, sc.image_size, sc.image_size), dtype=np.uint8)).to(dev)
        prop = rng.standard_normal((n, 10)).astype(np.float32)
        chunks = pol.rollout.step(frames, pol.dev.clip.text_tokens_cached(text), torch.from_numpy(prop).to(dev)).cpu()
        single = model.forward_inference(frames[:2], text[0], torch.from_numpy(prop[0]).to(dev)).cpu()
        return chunks, single

    for _ in range(2):
        pol.refresh()
        chunks, single = call()
        assert (chunks[0] - single).abs().max().item() <= 2e-4
    old_dev = pol.dev
    with torch.no_grad():
        model.action_head.mlp[2].bias.add_(0.5)
    pol.refresh()
    assert pol.dev is not old_dev and list(pol.rollout.lens_host) == [2, 2] and not pol.rollout.kv_ok.any()
    chunks, single = call()
    assert model.cache_size == 3 and list(pol.rollout.lens_host) == [3, 3]
    assert (chunks[0] - single).abs().max().item() <= 2e-4              # new head, old history: same as the single-episode path
    assert pol.rollout.last_mode == "recompute" and pol.rollout.kv_ok.all()
    pol.refresh()
    assert pol.rollout.step.__self__ is pol.rollout                     # unchanged weights: nothing is rebuilt
