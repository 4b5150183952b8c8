#!/bin/bash
# BASELINE config 5's ladder on one node with 8 MI355X: weak scaling 33/34/35/36 qubits on 1/2/4/8 GPUs
# (2^33 amplitudes = 128 GiB per GPU) and strong scaling of the 33-qubit QFT on 1/2/4/8 GPUs.  One JSON line per run;
# every multi-rank line carries parity_max_abs (sampled amplitudes vs the closed form), rccl_ranks and exchange_verified,
# and a run whose amplitudes are off exits 3 with an "error" line instead of a number.
#   usage: tools/scale_ladder.sh [OUTDIR] [STEPS] [WARMUP]
cd "$(dirname "$0")/.." || exit 1
out=${1:-gpurun_out/scale_ladder}
steps=${2:-5}
warm=${3:-2}
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
port=29611
run() {   # run NAME NGPUS QUBITS
  local name=$1 n=$2 q=$3
  port=$((port + 1))
  if [ "$n" -eq 1 ]; then
    python bench.py --gpus 1 --qubits "$q" --steps "$steps" --warmup "$warm" --no-configs --no-ladder-base --no-cpu-baseline \
      > "$out/$name.json" 2> "$out/$name.err"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port "$port" \
      bench.py --gpus "$n" --qubits "$q" --steps "$steps" --warmup "$warm" > "$out/$name.json" 2> "$out/$name.err"
  fi
  echo "$name rc=$? $(grep -o '"value": [^,]*' "$out/$name.json" | head -1) $(grep -o '"parity_max_abs": [^,]*' "$out/$name.json")"
}
ngpu=$(python - <<'PY'
from qcc_amd import native
print(native.device_count())
PY
)
echo "devices visible: $ngpu"
for n in 1 2 4 8; do
  [ "$n" -le "$ngpu" ] || continue
  g=0; m=$n; while [ "$m" -gt 1 ]; do m=$((m / 2)); g=$((g + 1)); done
  run "weak_q$((33 + g))_n$n" "$n" $((33 + g))
  run "strong_q33_n$n" "$n" 33
done
# one summary: measured vs predicted, weak / strong efficiency, parity_max_abs, rccl_ranks, exchange_verified
python tools/scale_summary.py "$out" > "$out/summary.json" && echo "summary: $out/summary.json"
