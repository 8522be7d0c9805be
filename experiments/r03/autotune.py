"""Random search over COMBINATIONS of the launch tunables through the default bench (the one-at-a-time sweeps found nothing):
python experiments/r03/autotune.py <n_trials> [steps]   -> prints every trial and the best ten."""
import json, os, random, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n, steps = int(sys.argv[1]), (sys.argv[2] if len(sys.argv) > 2 else "300")
space = {"SAGE_G_PER_CU": [5, 6, 7, 8], "SAGE_SO_THREADS": [256, 512, 1024], "SAGE_T16_GRID": [256, 384, 512, 768], "SAGE_T16_WAVES": [8, 16],
         "SAGE_DENSE_BLOCKS": [160, 192, 224, 256], "SAGE_DEPTH": [4, 5, 6], "SAGE_G_TRIP": [8, 16]}
rng = random.Random(7)
def run(env):
    e = dict(os.environ, **{k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", steps, "--warmup", "50", "--cpu-seconds", "0", "--no-variant", "--no-parity"],
                       env=e, capture_output=True, text=True, timeout=300)
    if r.returncode != 0:
        return None
    return json.loads(r.stdout.strip().splitlines()[-1])["ms_per_step"] * 1e3
res = []
base = {}
for t in range(n):
    env = base if t % 10 == 0 else {k: rng.choice(v) for k, v in space.items()}
    us = run(env)
    res.append((us if us is not None else 1e9, env))
    print(f"trial {t:3d}: {us if us is None else round(us, 2)} us  {env or 'DEFAULTS'}", flush=True)
res.sort(key=lambda x: x[0])
print("best ten:")
for us, env in res[:10]:
    print(f"  {us:.2f} us  {env or 'DEFAULTS'}")
d = [us for us, env in res if not env]
print(f"defaults: {len(d)} runs, {min(d):.2f} ... {max(d):.2f} us")
