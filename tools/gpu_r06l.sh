# round 6: how many calls in flight does the reference's chunk loop want?  (NEO360_LANES = side streams / scratch lanes of CallOverlap)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06l; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for n in 2 3 2 3; do
  NEO360_LANES=$n timeout 300 python bench.py --steps 4 --warmup 1 --cpu-rays 0 --exact-f32 0 --setup-timing 0 > $O/lanes$n.json 2> $O/lanes$n.err
  python - $n <<'PY'
import json, sys
d = json.load(open("gpurun_out/r06l/lanes%s.json" % sys.argv[1]))
c = d["chunk_loop"]
print("lanes", sys.argv[1], "headline", round(d["value"]), "chunk_loop", round(c["value"]), round(c["frac_of_headline"], 4), "bitwise", c["bitwise_equal_to_whole_frame_call"],
      {k: (round(v["value"]), round(v["chunk_loop"]["value"]), round(v["chunk_loop"]["frac_of_frame_call"], 3)) for k, v in d["other_workloads"].items()})
PY
done
