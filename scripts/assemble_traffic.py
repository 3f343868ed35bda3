#!/usr/bin/env python3
"""profiles/hbm_traffic.json from the *_pmc.json files of scripts/collect_profiles.py: per-launch HBM-side bytes of the apply
kernels, keyed by method, slices and (for the second workgroup shape of a plan) tile, which bench.py quotes as roofline.traffic
when the launch it timed has that shape.
usage: python scripts/assemble_traffic.py <tag> <name>=<key>:<tileW>x<tileH> ...   e.g. r02b bilinear_nz200=bilinear_nz200:512x8"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    recs = json.load(open(path)) if os.path.exists(path) else {}
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    for item in sys.argv[2:]:
        name, rest = item.split("=")
        key, tile = rest.split(":")
        src = "profiles/%s_%s_pmc.json" % (tag, name)
        p = json.load(open(os.path.join(ROOT, src)))
        recs[key] = {"bytes": p["hbm_bytes_per_launch"], "fetch_bytes": p["fetch_bytes_per_launch"], "write_bytes": p["write_bytes_per_launch"],
                     "tile": [int(v) for v in tile.split("x")], "kernel": p.get("kernel", ""),
                     "source": "%s @ %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 gfx950 correction, "
                               "calibrated in profiles/calib/)" % (src, head)}
    json.dump(recs, open(path, "w"), indent=1)
    print(json.dumps({k: [v["bytes"], v["tile"]] for k, v in recs.items()}, indent=1))


if __name__ == "__main__":
    main()
