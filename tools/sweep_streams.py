"""The headline under stream arrangements x idle streams created in front of the library's own (BHR_STREAM_PAD): one fresh
process per point.  usage: python tools/sweep_streams.py "BHR_HYBRID_SWAP=1" "BHR_HYBRID_SWAP=0 BHR_AUX_STREAMS=-1,1" ... [--pads a,b,c;a,b,c...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PADS = ["0,0,0", "1,0,0", "0,1,0", "0,0,1", "2,1,0", "1,1,1", "3,0,1", "0,2,2"]


def main():
    args = sys.argv[1:]
    pads = PADS
    if "--pads" in args:
        k = args.index("--pads")
        pads = args[k + 1].split(";")
        args = args[:k] + args[k + 2:]
    extra = []
    if "--bench" in args:
        k = args.index("--bench")
        extra = args[k + 1].split()
        args = args[:k] + args[k + 2:]
    out = {}
    for envs in args or [""]:
        fps, maps = [], []
        for pad in pads:
            env = dict(os.environ, BHR_STREAM_PAD=pad)
            for kv in envs.split():
                a, b = kv.split("=", 1)
                env[a] = b
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-other-math", "--no-config2", "--tile-workload", "none",
                                "--video-frames", "0"] + extra, env=env, capture_output=True, text=True)
            try:
                d = json.loads(p.stdout.strip().splitlines()[-1])
                fps.append(round(d["fps"], 1))
                maps.append("".join("S" if v else "." for v in d.get("stream_map", {}).values()) + " " + str(d.get("stream_calibration", {}).get("candidates_fps")))
            except Exception:
                fps.append(None)
                print(p.stderr[-400:], flush=True)
        ok = [f for f in fps if f]
        out[envs] = {"pads": pads, "fps": fps, "min": min(ok), "max": max(ok), "spread": round(max(ok) / min(ok) - 1, 4)}
        print("   maps (scene+slot0 scene+slot1 [scene+aux0 scene+aux1] slot0+slot1 ...):", maps, list(d.get("stream_map", {}).keys()) if ok else "")
        print(f"{envs or '(default)':60s} min {min(ok):7.1f} max {max(ok):7.1f} spread {100 * (max(ok) / min(ok) - 1):4.1f} %  {fps}", flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "stream_sweep.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
