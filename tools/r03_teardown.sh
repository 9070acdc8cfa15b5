# What a GPU process leaves behind for the next one: 6 identical processes back to back per variant, each prints its own hsa_init time
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o /tmp/r03_teardown tools/src/r03_teardown.cpp -lhsa-runtime64 || exit 1
for what in 0 1 2 3 4; do for how in 0 1 2; do
  sleep 2
  echo "== creates $what (0 nothing, 1 two streams, 2 + 4 GiB device memory, 3 + 96 MiB pinned, 4 both), ends $how (0 return, 1 _exit, 2 free then _exit)"
  for i in 1 2 3 4 5 6; do a=$(date +%s%N); out=$(timeout -k 5 60 /tmp/r03_teardown $what $how); b=$(date +%s%N); echo "$out   wall $(( (b - a) / 1000000 )) ms"; done
done; done
