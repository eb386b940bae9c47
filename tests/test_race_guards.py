"""Race / determinism guards for the hand-offs that deliberately bypass the language memory model's release / acquire pair.

Three places pass data between workgroups of ONE launch through write-through (sc1) stores, a drained store queue, a workgroup barrier, one
relaxed agent-scope arrival and sc1 loads in the last arriver (the model's fences measured 2x on these kernels, profiles/r04_a_*):
  * weight_grad_assemble_kernel's per-tile slab sums (ppo_dw2.hpp: the last of a tile's 4 row splits adds the slabs),
  * norm_batch_kernel's chunk moments and the reward job's discounted returns (ppo_kernels.hpp, NB_ST / NB_LD),
  * the narrow path's partial gradient vectors and the cooperative rollout's step words (ppo_narrow.hpp),
and the data-parallel peer exchange passes slots between PROCESSES with system-scope stores and flags (ppo_peer.hpp).  Every sum on these
paths has a fixed order, so a correct run is a pure function of its inputs: the tests below run the same workload several times -- hipGraph
replay, again, and with eager launches (different timing between the kernels) -- and require every weight, Adam slot, loss row and running
statistic to come out with the SAME BITS.  A stale or torn read in any hand-off shows up as a bitwise mismatch between two runs (it would move a
gradient tile or a chunk moment by a whole partial sum, not by rounding); the reference has no counterpart (one thread issues one
Session::Run per train step, ppo2/ppo2.hpp:430-468).  Iteration counts are sized for a few seconds per test; tools/soak*.py run the same
bodies for as long as one likes."""
import os
import subprocess
import sys

import numpy as np
import pytest

import ppo_cpp_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LR, CR, GAMMA, LAM = 3.93141e-4, 0.161023, 0.99, 0.95


def _run(monkeypatch, obs, act, hidden, E, T, nmb, epochs, iters, eager, bf16=False):
    if eager:
        monkeypatch.setenv("PPO_HIP_NO_GRAPH", "1")
    else:
        monkeypatch.delenv("PPO_HIP_NO_GRAPH", raising=False)
    g = ppo_cpp_amd.PPOHip(obs, act, list(hidden), compute_dtype=1 if bf16 else 0)
    g.init_orthogonal(0); g.norm_init(E, GAMMA); g.rollout_alloc(E, T)
    means = []
    for i in range(iters):
        g.collect_synthetic(1234, GAMMA, LAM, None, env0=0, step0=i * T, first=(i == 0))
        means.append(g.update(LR, CR, epochs, nmb, None, seed=1000 + i, want_rows=False)[1].copy())
    out = {"theta": g.get_flat(0), "adam_m": g.get_flat(1), "adam_v": g.get_flat(2), "means": np.array(means), "returns": g.rollout_get("returns")}
    for which, nm in ((0, "obs"), (1, "ret")):
        m, v, c = g.norm_stats(which)
        out[nm + "_mean"], out[nm + "_var"], out[nm + "_count"] = m, v, np.float64(c)
    counts = g.kernel_counts()
    g.close()
    monkeypatch.delenv("PPO_HIP_NO_GRAPH", raising=False)
    return out, counts


def _same_bits(a, b, what):
    assert np.isfinite(a["theta"]).all(), what
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg="%s: %s differs between two runs of the same workload" % (what, k))


def test_headline_shape_tile_and_statistics_hand_offs(monkeypatch):
    """BASELINE configs[2] exactly (4096 envs x 16 steps, [256,256], 32 minibatches x 10 epochs), 40 collect + update iterations = 12 800 train
    steps x 64 tile hand-offs (weight_grad_assemble_kernel) and 640 statistics launches per run: hipGraph replay twice and eager launches once."""
    a, counts = _run(monkeypatch, 18, 18, (256, 256), 4096, 16, 32, 10, 40, eager=False)
    assert counts.get("weight_grad_assemble_kernel", 0) > 0 and counts.get("train8_kernel", 0) > 0, counts
    b, _ = _run(monkeypatch, 18, 18, (256, 256), 4096, 16, 32, 10, 40, eager=False)
    c, _ = _run(monkeypatch, 18, 18, (256, 256), 4096, 16, 32, 10, 40, eager=True)
    _same_bits(a, b, "graph replay, run vs run")
    _same_bits(a, c, "graph replay vs eager launches")
    assert np.abs(a["means"][0] - a["means"][-1]).max() > 0             # (the run trained: the losses moved)


@pytest.mark.parametrize("E,T", [(5000, 24), (4097, 24), (150000, 2)])
def test_statistics_hand_off_at_row_counts_that_do_not_fill_lines(monkeypatch, E, T):
    """norm_batch_kernel at environment counts whose chunks do not end on 128-byte lines (5000 rows used to give 1000-row reward chunks; 4097 leaves
    a one-row chunk): the reward chunks are now whole multiples of 32 rows and every chunk's partial set sits on lines of its own, so the last
    arriver reads no line another workgroup shared with it.  30 rollouts (720 statistics launches) twice on the launch-per-kernel path: same bits;
    the first rollout's statistics against the oracle's two-pass moments."""
    from oracle import oracle as o
    monkeypatch.setenv("PPO_HIP_NO_PERSISTENT_COLLECT", "1")
    iters = 30 if E < 100000 else 6
    runs = []
    for _ in range(2):
        g = ppo_cpp_amd.PPOHip(18, 18, [64, 64]); g.init_orthogonal(0); g.norm_init(E, GAMMA); g.rollout_alloc(E, T)
        first = None
        for i in range(iters):
            g.collect_synthetic(1234, GAMMA, LAM, None, env0=0, step0=i * T, first=(i == 0))
            if i == 0:
                first = (g.norm_stats(0), g.norm_stats(1), g.rollout_get("rewards"))
        out = {"returns": g.rollout_get("returns"), "rewards": g.rollout_get("rewards"), "obs": g.rollout_get("obs")}
        for which, nm in ((0, "obs"), (1, "ret")):
            m, v, c = g.norm_stats(which)
            out[nm + "_mean"], out[nm + "_var"], out[nm + "_count"] = m, v, np.float64(c)
        kc = g.kernel_counts()                                           # the launch-per-kernel path: norm_batch_kernel per env step, no resident rollout
        assert kc["narrow_rollout_coop_kernel"] == 0 and kc["narrow_rollout_kernel"] == 0 and kc["narrow_collect_kernel"] == 0, kc
        g.close()
        runs.append((out, first))
    for k in runs[0][0]:
        np.testing.assert_array_equal(runs[0][0][k], runs[1][0][k], err_msg=k)
    if E <= 5000:                                                        # (the scalar oracle's rollout: seconds at these sizes)
        orc = o.Oracle(18, 18, [64, 64]); orc.init_orthogonal(0)
        nz = o.Normalizer(E, 18)
        noise = np.zeros((T, E, 18), np.float32)                        # the statistics do not depend on the actions of the seeded env
        ro, _, _ = o.collect(orc, nz, 1234, T, noise, GAMMA, LAM)
        (m, v, c), (rm, rv, rc), rew = runs[0][1]
        np.testing.assert_allclose(m, nz.obs_rms.mean, rtol=1e-5, atol=1e-6); np.testing.assert_allclose(v, nz.obs_rms.var, rtol=1e-5)
        np.testing.assert_allclose(rv, nz.ret_rms.var, rtol=1e-5); assert c == nz.obs_rms.count and rc == nz.ret_rms.count
        np.testing.assert_allclose(rew, ro["rewards"], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("E,T,nmb,iters", [(1024, 64, 32, 100), (48, 32, 4, 300), (1, 512, 8, 150), (1, 2048, 32, 30)])
def test_narrow_path_partial_vectors_and_cooperative_rollout(monkeypatch, E, T, nmb, iters):
    """The reference's [64,64] shape: narrow_train_kernel's per-workgroup partial gradient vectors (write-through, summed by narrow_reduce_kernel),
    the deferred Adam reading the other parameter set, and -- at 1024 environments, BASELINE configs[3]'s count -- the cooperative persistent rollout
    whose workgroups meet once per env step through step words: graph replay twice, eager once.  With minibatches of <= 64 rows (the last two cases; the
    last one is the reference's own command line, 1 environment x 2048 steps x 32 minibatches) the epoch runs inside narrow_epoch_kernel, whose four
    workgroups exchange their partial vectors through one XCD's L2 with ordinary stores and loads and meet once per minibatch: 4800 / 3840 meetings per run."""
    a, counts = _run(monkeypatch, 18, 18, (64, 64), E, T, nmb, 4, iters, eager=False)
    assert counts["narrow_epoch_kernel" if E * T // nmb <= 64 else "narrow_train_kernel<static>"] > 0, counts
    b, _ = _run(monkeypatch, 18, 18, (64, 64), E, T, nmb, 4, iters, eager=False)
    c, _ = _run(monkeypatch, 18, 18, (64, 64), E, T, nmb, 4, iters, eager=True)
    _same_bits(a, b, "graph replay, run vs run")
    _same_bits(a, c, "graph replay vs eager launches")


def test_bf16_path_slabs_and_split_k(monkeypatch):
    """configs[4]'s shape (256 / 64 / [1024]^3, bf16): the work-balanced weight-gradient GEMM's 2-3 partial slabs per tile, the split-K heads
    and the column-sum slots are all added in a fixed order: 4 iterations of 2 epochs x 8 minibatches of 4096 rows, graph replay twice + eager."""
    a, _ = _run(monkeypatch, 256, 64, (1024, 1024, 1024), 2048, 16, 8, 2, 4, eager=False, bf16=True)
    b, _ = _run(monkeypatch, 256, 64, (1024, 1024, 1024), 2048, 16, 8, 2, 4, eager=False, bf16=True)
    c, _ = _run(monkeypatch, 256, 64, (1024, 1024, 1024), 2048, 16, 8, 2, 4, eager=True, bf16=True)
    _same_bits(a, b, "graph replay, run vs run")
    _same_bits(a, c, "graph replay vs eager launches")


@pytest.mark.parametrize("world,hidden,E,T,nmb,epochs,iters", [(8, (256, 256), 64, 8, 4, 4, 12), pytest.param(8, (64, 64), 128, 16, 4, 4, 12, marks=pytest.mark.slow)])
def test_peer_exchange_between_eight_processes_is_deterministic(tmp_path, world, hidden, E, T, nmb, epochs, iters):
    """World 8 on the one-shot peer path (slots pushed into every peer's region, a flag per source, rank-ordered sums: ppo_peer.hpp; the statistics
    table pushed by norm_batch_kernel's last workgroups): 12 collect + update iterations = 192 gradient exchanges, 48 advantage-moment exchanges
    and ~100 statistics exchanges between 8 processes, run TWICE: every rank of both runs must hold the same bits (a slot read before its flag's
    data had landed, or a parity slot reused too early, would change a whole rank's contribution)."""
    from tests.test_dp_two_ranks import build_fake_rccl, run_workers
    fake = build_fake_rccl(str(tmp_path))
    import ppo_cpp_amd as pk                                             # noqa: F401  (the workers import the same build)
    from oracle import oracle as o
    orc = o.Oracle(18, 18, list(hidden)); orc.init_orthogonal(21)
    res = []
    for run in range(2):
        sub = os.path.join(str(tmp_path), "run%d" % run); os.makedirs(sub)
        uid = np.zeros(128, np.uint8)
        name = ("/ppo_dp_soak_%d_%d" % (os.getpid(), run)).encode()
        uid[:len(name)] = np.frombuffer(name, np.uint8)
        fin = os.path.join(sub, "in.npz")
        np.savez(fin, hidden=np.array(hidden), E=E, T=T, nmb=nmb, epochs=epochs, theta=orc.theta, uid=uid, gamma=GAMMA, lam=LAM, seed=1234, lr=LR, cr=CR)
        env = dict(os.environ, PPO_RCCL_LIBRARY=fake, HSA_ENABLE_IPC_MODE_LEGACY="0", PPO_TEST_PEER="1", PPO_TEST_ITERS=str(iters))
        res.append(run_workers(sub, world, fin, env))
    keys = ("means", "theta", "adam_m", "adam_v", "obs_mean", "obs_var", "obs_count", "ret_mean", "ret_var", "ret_count")
    for out in res[0] + res[1]:
        assert int(out["peer"]) == 1
        for k in keys:
            np.testing.assert_array_equal(res[0][0][k], out[k], err_msg=k)          # replicas of one run AND the two runs: one set of bits
    for r in range(world):
        np.testing.assert_array_equal(res[0][r]["ro_returns"], res[1][r]["ro_returns"])
    assert np.isfinite(res[0][0]["theta"]).all() and np.abs(res[0][0]["theta"] - orc.theta).max() > 0


@pytest.mark.parametrize("hidden,E,T", [((256, 256), 4096, 48), ((256, 256), 4093, 24), ((128, 96), 77, 40), ((256, 256), 1, 40)])
def test_actions_published_by_the_policy_kernel_are_the_rollout_rows(monkeypatch, hidden, E, T):
    """Host-Env path for nets wider than 64 (ppo_rollout_act, reference ppo2/runner.hpp:75-116): the policy tower's workgroups store their 16 rows of
    actions straight into pinned memory and raise one word each; the host copies a block out as soon as its word shows the call's sequence number.
    No fence, no stream synchronisation: what orders the bytes before the word is `s_waitcnt vmcnt(0)` on uncached stores.  A block read before
    its bytes had landed would differ from the rollout buffer's row: every env step's returned actions must be the SAME BITS as rollout field
    `actions`, over T steps x several rollouts, at a block count of 256, at ragged last blocks (4093: 13 rows; 77; 1), and equal to the
    copy-engine form (PPO_HIP_NO_DIRECT_ACT=1) including the fields the still-running value tower writes."""
    # (the library uses this form up to 64 environments -- beyond that the copy engine is faster; the override lets the test run it at 256 blocks)
    monkeypatch.setenv("PPO_HIP_DIRECT_ACT_MAX_BLOCKS", "4096")
    rng = np.random.RandomState(5)
    trans = [(rng.uniform(-1, 1, (E, 18)).astype(np.float32), rng.uniform(-1, 1, E).astype(np.float32), (rng.uniform(size=E) < 0.05).astype(np.float32))
             for _ in range(2 * T + 1)]
    outs = {}
    for form in ("direct", "copy"):
        if form == "copy":
            monkeypatch.setenv("PPO_HIP_NO_DIRECT_ACT", "1")
        else:
            monkeypatch.delenv("PPO_HIP_NO_DIRECT_ACT", raising=False)
        g = ppo_cpp_amd.PPOHip(18, 18, list(hidden)); g.init_orthogonal(2); g.seed(7); g.norm_init(E, GAMMA); g.rollout_alloc(E, T)
        g.rollout_reset(trans[0][0])
        got = []
        k = 1
        for ro in range(2):
            acts = []
            for t in range(T):
                a = g.rollout_act(t, None)
                acts.append(a.copy())
                g.rollout_observe(t, *trans[k]); k += 1
            g.rollout_finish(GAMMA, LAM)
            np.testing.assert_array_equal(np.stack(acts), g.rollout_get("actions"), err_msg="%s: returned actions vs rollout rows (rollout %d)" % (form, ro))
            got.append({f: g.rollout_get(f) for f in ("obs", "actions", "values", "neglogp", "rewards", "returns", "dones")})
        g.close()
        outs[form] = got
    monkeypatch.delenv("PPO_HIP_NO_DIRECT_ACT", raising=False)
    for a, b in zip(outs["direct"], outs["copy"]):
        for f in a:
            np.testing.assert_array_equal(a[f], b[f], err_msg=f)


_GRAPH_SHAPES = [((64, 64), 18, 18, 1, 512, 8, False), ((64, 64), 18, 18, 64, 64, 4, False), ((256, 256), 18, 18, 64, 16, 4, False), ((256, 256), 36, 18, 512, 16, 4, False),
                 ((512, 256, 256), 18, 18, 32, 16, 4, False), ((1024, 1024, 1024), 256, 64, 256, 16, 4, True)]


@pytest.mark.parametrize("hidden,O,A,E,T,nmb,bf16", _GRAPH_SHAPES)
def test_update_graph_holds_kernel_nodes_only(hidden, O, A, E, T, nmb, bf16):
    """The rule behind round 6's finding (ppo_hip.hip, zero_words): the launch sequence ppo_update captures and replays holds KERNEL nodes only.  A hipMemsetAsync node at its
    head (weight_grad_assemble_kernel's arrival counters) replayed out of order on ROCm 7.0.2 in a long-lived process and left every [256,256] update wrong by 1e-2 from then
    on; memset / memcpy nodes are therefore not allowed in the graph at all.  Every kernel family's update (resident epoch kernel, narrow launches, the [256,256] pair, the
    round-2 kernels behind three hidden layers, the bf16 path) with the on-device shuffle and with explicit permutations."""
    g = ppo_cpp_amd.PPOHip(O, A, list(hidden), compute_dtype=1 if bf16 else 0)
    g.init_orthogonal(0); g.norm_init(E, GAMMA); g.rollout_alloc(E, T)
    g.collect_synthetic(7, GAMMA, LAM, None, env0=0, step0=0, first=True)
    for explicit in (False, True):
        perms = np.stack([np.random.RandomState(e).permutation(E * T).astype(np.int32) for e in range(2)]) if explicit else None
        g.update(LR, CR, 2, nmb, perms, seed=1, want_rows=False)
        nodes = g.debug_graph_nodes()
        assert nodes is not None and nodes["kernel"] >= 2 * nmb // 8 + 1, nodes
        assert nodes["memset"] == 0 and nodes["memcpy"] == 0 and nodes["other"] == 0, "the update's graph must hold kernel nodes only: %s" % nodes
    g.close()


@pytest.mark.parametrize("hidden,O,A,E,T,nmb,bf16", _GRAPH_SHAPES)
def test_results_do_not_depend_on_what_the_lds_held(hidden, O, A, E, T, nmb, bf16):
    """Every kernel must write the LDS it reads: two collect + update iterations with a NaN pattern left in every LDS word of every CU in front of each call
    (ppo_debug_poison_lds) against the same run without -- same bits.  (A workgroup that reads LDS it never wrote sees what the previous workgroup on its CU left there: a
    result that depends on which kernels ran before, which is what round 5's open finding looked like before its cause was known.)"""
    outs = []
    for poison in (False, True):
        g = ppo_cpp_amd.PPOHip(O, A, list(hidden), compute_dtype=1 if bf16 else 0)
        g.init_orthogonal(0); g.norm_init(E, GAMMA); g.rollout_alloc(E, T)
        acc = []
        for it in range(2):
            if poison:
                g.debug_poison_lds()
            g.collect_synthetic(7, GAMMA, LAM, None, env0=0, step0=it * T, first=(it == 0))
            acc += [g.rollout_get(f) for f in ("actions", "values", "neglogp", "returns")]
            if poison:
                g.debug_poison_lds()
            rows, mean = g.update(LR, CR, 2, nmb, None, seed=3 + it)
            acc += [rows.copy(), g.get_flat(0), g.get_flat(1), g.get_flat(2)]
        assert np.isfinite(acc[-3]).all()
        outs.append(acc)
        g.close()
    for k, (a, b) in enumerate(zip(*outs)):
        np.testing.assert_array_equal(a, b, err_msg="output %d differs once the LDS holds NaNs in front of every call" % k)


def test_graph_replays_under_the_hip_runtime_bundled_with_torch(tmp_path):
    """The TRIGGER of round 5's open finding, in a fresh process and under a minute: the process imports torch first, so the HIP runtime bundled with the PyTorch wheel
    (same soname as the system's, an older release) serves libppo_hip.so -- what happens to every pytest run that collects a module importing torch.  On that runtime the
    memset node rounds 3 - 5 kept at the head of the update's captured graph replayed out of order from the first or second replay on (arrival counters non-zero behind the
    update, weights off by 4e-3: profiles/r06_a_*); with kernel nodes only the [256,256] handle's six updates -- five replays -- match the oracle beside the narrow handle
    and alone, and the counters are zero behind every one.  (Where the wheel bundles no runtime of its own the test still holds: it then runs on the system's.)"""
    import json
    out = os.path.join(str(tmp_path), "res.json")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "torch_runtime_worker.py"), out], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    res = json.load(open(out))
    assert len(res["hip_runtimes"]) == 1, res["hip_runtimes"]                 # one runtime serves the process
    for name in ("beside the narrow handle", "alone"):
        assert res[name]["nonzero_counters"] == [0] * 6, (name, res)
        assert max(res[name]["weight_err"]) < 5e-6, (name, res)
    # ... and the bf16 path's event fork / join inside the captured graph (bucketed exchange, one-rank communicator over the real collective library) on the same runtime
    assert res["bucketed_vs_single_theta_maxdiff"] < 1e-4 and res["bucketed_vs_single_rows_maxdiff"] < 5e-2, res
    print("HIP runtime in the worker:", res["hip_runtimes"], "| collectives captured:", res["bucketed 2 graph collectives"])
