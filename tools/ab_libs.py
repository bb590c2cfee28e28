"""Same-box A/B of library variants in ONE gpurun call: for every library (built with tools/gpu/build_variant.sh, or
"default" for audioldm2_amd/libaldm_hip.so) run a parity subset of the op tests and the graph-replayed UNet step probe,
alternating the step probes so drift of the box shows up as spread rather than as a difference.
Usage: python tools/ab_libs.py [--model NAME] [--reps N] [--tests "pytest -k expression"] default tools/gpu/libaldm_x.so ...
Environment per variant can be given as  path::VAR=VALUE,VAR2=VALUE2  (e.g. default::ALDM_ATTN_MMA=bf16x6)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    model, reps, tests = "audioldm2-full", 2, "igemm or conv or linear or attention or groupnorm"
    while args and args[0].startswith("--"):
        k, v = args[0], args[1]
        args = args[2:]
        if k == "--model":
            model = v
        elif k == "--reps":
            reps = int(v)
        elif k == "--tests":
            tests = v
    variants = []
    for a in args or ["default"]:
        path, _, envs = a.partition("::")
        env = dict(os.environ)
        if path != "default":
            env["ALDM_LIB_PATH"] = os.path.join(ROOT, path) if not os.path.isabs(path) else path
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("=")
            env[k] = v
        variants.append((a, env))
    for name, env in variants:
        if tests and tests != "none":
            r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_ops_gpu.py"), "-x", "-q", "-m",
                                "gpu", "-k", tests], env=env, capture_output=True, text=True, cwd=ROOT)
            print(f"[{name}] op parity: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
    for rep in range(reps):
        for name, env in variants:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_probe.py"), model, "1"], env=env,
                               capture_output=True, text=True, cwd=ROOT)
            line = [l for l in r.stdout.splitlines() if "unet step" in l]
            print(f"[{name}] rep {rep}: {line[-1] if line else 'FAILED ' + r.stderr[-300:]}", flush=True)


if __name__ == "__main__":
    main()
