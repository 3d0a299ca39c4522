#!/bin/bash
# Timing-only ablations of the headline encode kernel (k_encode_mfma_lds3<8,20,true,u8>), each as its OWN library
# reductive_amd/libpqhip_timing_<n>.so (never the shipped one): ms per launch, in-kernel clock and cycles per tile.
#   tools/enc_power_ab.sh build "0 3 5 6"   (here, cross-compiles)       tools/enc_power_ab.sh run "0 3 5 6" <tag>   (GPU box)
set -e
cd "$(dirname "$0")/.."
mode=$1; variants=${2:-"0 3 5 6"}; tag=${3:-encab}
C=reductive_amd/csrc
if [ "$mode" = build ]; then
  make -s -C $C -j8
  mkdir -p $C/build_timing
  for n in $variants; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Iinclude \
        -DPQ_KIND=2 -DPQ_T=8 -DPQ_DPSET=0 -DPQHIP_TIMING_ONLY_BUILD -DENC_ABLATE=$n ${EXTRA_DEFS} -c $C/encode_launch.hip -o $C/build_timing/enc_abl_$n.o &
  done
  wait
  for n in $variants; do
    objs=$(ls $C/build/*.o | grep -v encode_launch_k2_t8_p0.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC $objs $C/build_timing/enc_abl_$n.o -o reductive_amd/libpqhip_timing_$n.so -shared -Wl,-rpath,/opt/rocm/lib -lpthread
  done
  ls -la reductive_amd/libpqhip_timing_*.so
  exit 0
fi
out=gpurun_out/$tag; mkdir -p $out
for round in 1 2; do
  for n in $variants; do
    echo "== variant $n round $round" | tee -a $out/log.txt
    PQHIP_LIB=$PWD/reductive_amd/libpqhip_timing_$n.so python bench.py --workload encode --no-cpu-baseline --no-sub-configs --steps 20 --warmup 5 2>/dev/null \
      | python -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  ms_per_step', r['ms_per_step'], 'frac', r['roofline']['frac'])" | tee -a $out/log.txt
    PQHIP_LIB=$PWD/reductive_amd/libpqhip_timing_$n.so PQHIP_DEBUG_ENC_STAMP=1 python bench.py --workload encode --no-cpu-baseline --no-sub-configs --steps 3 --warmup 2 2>&1 \
      | grep "encode stamps" | tail -2 | tee -a $out/log.txt
  done
done
