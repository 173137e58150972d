"""Ad-hoc SQ counter passes over one bench.py step (run ON the GPU box):

    python tools/pmc_probe.py TAG "CTR_A CTR_B ..." ["CTR_C ..."] [-- bench args]

One rocprofv3 --pmc pass per quoted group (counters never together with traces); prints the mean
per launch of every counter for each engine kernel and writes gpurun_out/TAG/pmc_probe.json.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import profile_step as ps  # noqa: E402


def main():
    args = sys.argv[1:]
    bench_args = []
    if "--" in args:
        i = args.index("--")
        args, bench_args = args[:i], args[i + 1:]
    tag, groups = args[0], [g.split() for g in args[1:]]
    outdir = os.path.join(ps.ROOT, "gpurun_out", tag)
    os.makedirs(outdir, exist_ok=True)
    result = {}
    for n, group in enumerate(groups):
        db, _ = ps.run_pass(outdir, "g%d" % n, ["--pmc"] + group, bench_args)
        for name, m in ps.counter_means(db, set(group)).items():
            s = ps.short(name)
            if not s or "tpamd" not in name:
                continue
            for c, (calls, mean) in m.items():
                result.setdefault(s, {})[c] = mean
    for k, m in sorted(result.items()):
        print(k)
        for c, v in sorted(m.items()):
            print("   %-32s %16.1f" % (c, v))
    with open(os.path.join(outdir, "pmc_probe.json"), "w") as f:
        json.dump(result, f, indent=1)


if __name__ == "__main__":
    main()
