"""CPU legs of bench.py, both built on the stand-alone oracle (oracle/levels.py, one Python env object per environment
-- the reference's own execution model, babyai/rl/utils/penv.py:4-16).  TEST / REPORTING INFRASTRUCTURE: a reported
baseline (kind = "port") and the in-run parity checker, never a product path.

  run(...)            the same workload as the GPU leg (same level, same counter-based action stream keyed on the global
                      env index, auto-reset, optional pixel wrapper) on the usable host cores: one env per process on every
                      core + the single-core figure, with the parallel efficiency of the pool.
  parity_replay(...)  re-steps the first envs of a shard from their seeds through the recorded steps of a bench run and
                      compares EVERY output byte (image, direction, float64 reward bits, done, pixels) with what the
                      engine produced inside the timed region.
"""
import multiprocessing as mp
import os
import sys
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)


def usable_cores():
    """Cores this process may really use: the affinity mask, cut by a cgroup CPU quota when there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        quota = q / float(f.read().split()[0])
            break
        except (OSError, ValueError, IndexError):
            continue
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def make_pool(processes=None):
    """A worker pool forked NOW.  bench.py calls this before it initialises the GPU runtime / the process group: forking a
    process that already runs HIP and RCCL threads is asking for trouble, so the workers exist first and get their work
    (pickled slices of the recorded outputs) later."""
    return mp.get_context("fork").Pool(processes or usable_cores())


def _make(level, pixel, seed):
    if _ROOT not in sys.path:
        sys.path.insert(0, _ROOT)
    from oracle import levels as olevels
    from gym_minigrid.wrappers import RGBImgPartialObsWrapper
    env = olevels.make_env(level)
    env.seed(int(seed))
    return env, (RGBImgPartialObsWrapper(env) if pixel else env)


def _worker(args):
    level, pixel, seconds, seed_base, action_seed, index = args
    if _ROOT not in sys.path:
        sys.path.insert(0, _ROOT)
    from babyai_amd.action_stream import action_scalar
    _, wrapped = _make(level, pixel, seed_base + index)
    wrapped.reset()
    steps = 0
    t0 = time.perf_counter()
    while True:
        for _ in range(256):
            _, _, done, _ = wrapped.step(action_scalar(action_seed, steps, index))
            steps += 1
            if done:
                wrapped.reset()
        if time.perf_counter() - t0 >= seconds:
            break
    return steps, time.perf_counter() - t0


def run(level="BossLevel", pixel=True, seconds=12.0, seed_base=0, action_seed=1234, pool=None, cores=None):
    """`cores` = workers to use (default: every usable core; bench.py passes the size of the pool it forked -- on a
    multi-rank node rank 0 leaves two cores to each other rank)."""
    cores = cores or usable_cores()
    own = pool is None
    if own:
        pool = make_pool(cores)
    try:
        # (time-outs: a worker that dies must cost the bench a field, not hang it)
        one_steps, one_dt = pool.apply_async(_worker, ((level, pixel, min(4.0, seconds / 3), seed_base, action_seed, 0),)).get(timeout=120 + seconds)
        res = pool.map_async(_worker, [(level, pixel, seconds, seed_base, action_seed, i) for i in range(cores)], chunksize=1).get(timeout=180 + 3 * seconds)
    finally:
        if own:
            pool.terminate()
    total = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    single = one_steps / one_dt
    return {
        "value": total / wall, "unit": "env-steps/s", "cores": cores, "kind": "port",
        "sample": "oracle/levels.py BabyAI-%s-v0%s: envs 0..%d of the GPU leg's batch (same seeds, same counter-based action "
                  "stream), one process per usable core x %.0f s, auto-reset (%d steps)"
                  % (level, " + RGBImgPartialObsWrapper" if pixel else "", cores - 1, seconds, total),
        "single_core_value": single, "parallel_efficiency": (total / wall) / (cores * single),
        "os_cpu_count": os.cpu_count(), "usable_cores": usable_cores(),
    }


def reference_over_port(level, pixel):
    """The measured speed ratio reference / port for this workload, from profiles/r0N/cpu_port_vs_reference.json (taken in
    the build container, where /root/reference can be imported: tools/cpu_port_vs_reference.py -- same seeds, same actions,
    equal output digests).  The reference tree never reaches the GPU box, so the bench line carries the port's figure and
    this ratio to convert it.  {} when no measurement is on file for the workload."""
    import json
    table = row = rel = None
    for rnd in ("r04", "r03"):                              # the latest measurement on file
        rel = "profiles/%s/cpu_port_vs_reference.json" % rnd
        try:
            with open(os.path.join(_ROOT, rel)) as f:
                table = json.load(f)
            row = table["workloads"]["%s/%s" % (level, "pixel" if pixel else "encoded")]
            break
        except (OSError, ValueError, KeyError):
            row = None
    if row is None:
        return {}
    return {"reference_over_port": row["reference_over_port"],
            "reference_over_port_provenance": {"file": rel, "digest_equal": row["digest_equal"],
                                               "reference_steps_per_s": row["reference_steps_per_s"], "port_steps_per_s": row["port_steps_per_s"],
                                               "steps": row["steps"], "where": table.get("where"), "commit": table.get("commit")}}


# ---- in-run parity --------------------------------------------------------------------------------------------------


def _replay(args):
    """args carry a slice of the log: arrays [steps(+1), hi - lo, ...]; pix = how many of the slice's first envs have pixels"""
    level, pix, seed_base, action_seed, ids, log = args
    import numpy as np
    if _ROOT not in sys.path:
        sys.path.insert(0, _ROOT)
    from babyai_amd.action_stream import action_scalar
    steps = log["done"].shape[0]
    pixel_envs = pix
    bad, where = 0, None
    for k in range(len(ids)):
        gid = int(ids[k])                                   # global env index: seed and action stream key
        env, wrapped = _make(level, k < pixel_envs, seed_base + gid)
        o = wrapped.reset()
        enc = env.gen_obs() if k < pixel_envs else o

        def check(t, o, enc, r, d):
            nonlocal bad, where
            ok = np.array_equal(enc["image"], log["image"][t + 1, k]) and int(enc["direction"]) == int(log["direction"][t + 1, k])
            if t >= 0:
                ok = ok and np.float64(r).view(np.uint64) == log["reward64"][t, k].view(np.uint64) and bool(d) == bool(log["done"][t, k])
            if k < pixel_envs:
                ok = ok and np.array_equal(o["image"], log["pixels"][t + 1, k])
            if not ok:
                bad += 1
                if where is None:
                    where = {"env": gid, "step": int(t)}

        check(-1, o, enc, 0.0, False)
        for t in range(steps):
            o, r, d, _ = wrapped.step(action_scalar(action_seed, t, gid))
            if d:
                o = wrapped.reset()
            enc = env.gen_obs() if k < pixel_envs else o
            check(t, o, enc, r, d)
    return bad, where


def parity_replay(level, log, seed_base, action_seed, first, pixel_envs=0, env_ids=None, pool=None):
    """log: dict of numpy arrays recorded by the bench: image uint8[S+1, P, 7,7,3] (index 0 = after reset()), direction
    uint8[S+1, P], reward64 float64[S, P], done uint8[S, P], pixels uint8[S+1, pixel_envs, 56,56,3] -- the outputs of the
    shard's first P envs at every step (or of the global envs `env_ids`, in that order).  Returns {"envs", "steps", "mismatches", "first_mismatch", "seconds", "cores"}."""
    P = log["done"].shape[1]
    cores = usable_cores()
    t0 = time.perf_counter()
    nchunk = min(P, cores * 4)
    bounds = [P * c // nchunk for c in range(nchunk + 1)]
    ids = [first + k for k in range(P)] if env_ids is None else [int(i) for i in env_ids]
    assert len(ids) == P
    jobs = []
    for c in range(nchunk):
        lo, hi = bounds[c], bounds[c + 1]
        part = {k: (v[:, lo:hi] if k != "pixels" else v[:, lo:min(hi, pixel_envs)]) for k, v in log.items() if k != "pixels" or lo < pixel_envs}
        jobs.append((level, max(0, min(hi, pixel_envs) - lo), seed_base, action_seed, ids[lo:hi], part))
    own = pool is None
    if own:
        pool = make_pool(cores)
    try:
        res = pool.map_async(_replay, jobs, chunksize=1).get(timeout=900)
    finally:
        if own:
            pool.terminate()
    bad = sum(r[0] for r in res)
    firsts = [r[1] for r in res if r[1] is not None]
    return {"envs": P, "pixel_envs": int(pixel_envs), "steps": int(log["done"].shape[0]), "outputs": "image, direction, f64 reward bits, "
            "done every step" + (", pixels of the first %d envs" % pixel_envs if pixel_envs else ""),
            "mismatches": int(bad), "first_mismatch": firsts[0] if firsts else None,
            "oracle": "oracle/levels.py", "seconds": time.perf_counter() - t0, "cores": cores}


def c1(steps=10000, seed=0, use_reference=False):
    """BASELINE.json configs[0]: BabyAI-GoToRedBall-v0, ONE env, seed 0, `steps` random actions over
    all 7 actions with auto-reset, encoded obs -- the plumbing / single-core CPU figure.  `use_reference` runs the
    reference's own level class on the shim instead of the stand-alone oracle (build container only).
    Also returns a digest of every (obs, reward, done) so the two can be compared."""
    import hashlib
    if _ROOT not in sys.path:
        sys.path.insert(0, _ROOT)
    import numpy as np
    if use_reference:
        from oracle import refenv
        refenv.import_reference()
        import gym
        env = gym.make("BabyAI-GoToRedBall-v0")
    else:
        from oracle import levels as olevels
        env = olevels.make_env("GoToRedBall")
    env.seed(seed)
    obs = env.reset()
    acts = np.random.RandomState(seed).randint(0, 7, size=steps)
    h = hashlib.sha256()
    episodes = 0
    t0 = time.perf_counter()
    for a in acts:
        obs, reward, done, _ = env.step(int(a))
        h.update(obs["image"].tobytes())
        h.update(np.float32(reward).tobytes())
        h.update(bytes([int(obs["direction"]), int(done)]))
        if done:
            episodes += 1
            obs = env.reset()
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "reference" if use_reference else "port",
            "sample": "BabyAI-GoToRedBall-v0, 1 env, seed %d, %d random-action steps, %d episodes" % (seed, steps, episodes),
            "digest": h.hexdigest()}


if __name__ == "__main__":
    import json
    if sys.argv[1:2] == ["c1"]:
        print(json.dumps(c1(use_reference="--reference" in sys.argv)))
        sys.exit(0)
    print(json.dumps(run(*(sys.argv[1:2] or ["BossLevel"]), pixel="--no-pixel" not in sys.argv, seconds=4.0)))
