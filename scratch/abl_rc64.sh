# variants of csrc/policy_rc64.hip built on the GPU box (this file alone) and timed stand-alone at 65 536 rows
cd $GRAFT_REPO_ROOT
SRC=${RC64_SRC:-scratch/ubench/policy_rc64_experiment.hip.txt}
mkdir -p /tmp/rcv
build() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -x hip $2 $SRC -o /tmp/rcv/$1.so 2>/dev/null; }
run() { RC64_LIB=/tmp/rcv/$1.so RC64_TAG="$1" timeout -k 10 120 python scratch/bench_rc64.py 2>&1 | grep -v amdgpu.ids; }
for v in "$@"; do
  name=$(echo "$v" | tr ' =' '__' | tr -d '-'); [ -z "$name" ] && name=default
  build "$name" "$v" && run "$name" || echo "$name: build or run failed"
done
