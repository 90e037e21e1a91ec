# round 6: what made scene_setup_ms jump from 7.6 to 400 ms between two builder runs?  (NUMA binding / call overlap / box)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06j; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
one() { env $2 timeout 200 python bench.py --steps 2 --warmup 1 --cpu-rays 0 --others 0 --exact-f32 0 --chunk-loop 0 > $O/$1.json 2> $O/$1.err
  python - $1 <<'PY'
import json, sys
d = json.load(open("gpurun_out/r06j/%s.json" % sys.argv[1]))
print(sys.argv[1], "scene_setup_ms", round(d["scene_setup_ms"], 1), [round(x, 1) for x in d["scene_setup"]["runs_ms"]], "set_scene", round(d["scene_setup"]["set_scene_ms"], 2), "rays/s", round(d["value"]))
PY
}
one default "A=1"
one nobind "NEO360_NUMA_BIND=0"
one nooverlap "NEO360_OVERLAP=0"
one nobind_nooverlap "NEO360_NUMA_BIND=0 NEO360_OVERLAP=0"
one default2 "A=1"
