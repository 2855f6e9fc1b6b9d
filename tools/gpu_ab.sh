#!/bin/bash
# tools/gpu_ab.sh — the A/B runs of a GPU call, one script (everything alternates on ONE box: boxes differ by +-3 %, and so do runs).
#   gpu_ab.sh tests [files...]               the decoder's and the packer's GPU tests, default build (first thing in a call)
#   gpu_ab.sh inflate KINDS LIB...           decoder kernel times (tools/inflate_time.py) per file kind ("headline,hard,real") and build, twice
#   gpu_ab.sh bench REPS LIB...              the bench's headline legs per build
#   gpu_ab.sh env VAR REPS VALUE...          ... per value of an environment variable (GPU_MAX_HW_QUEUES, TCMI_TEAM_BYTES, ...)
#   gpu_ab.sh ctxopt KEY REPS VALUE...       ... per value of a context option (tcmi_ctx_set_option: mid_wait, prefix_kernels, ...)
#   gpu_ab.sh args REPS "ARGS"...            ... per set of bench flags ("--gpu-streams 3" "--gpu-streams 8": contexts; "--decoders 4 --gpu-streams 6")
#   gpu_ab.sh inflate-env VAR KINDS VALUE... decoder kernel times per value of an environment variable (TCMI_SYM_WINDOW, TCMI_SYM_BLOCKS, ...)
#   gpu_ab.sh stamps KINDS LIB...            phase clocks of bgzf_symbols / bgzf_copy (tools/inflate_stamps.py; builds with -DTCMI_COPY_PHASES for bgzf_copy's)
# LIB: "default" = trueconsense_amd/lib/libtcmi.so, else trueconsense_amd/lib/var/lib<LIB>.so (tools/build_variant.sh NAME "flags";
# libbase.so: a copy of the build before the change under test).  Extra bench flags: BENCH_ARGS="--gpu-streams 6".
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
mode=$1; shift
lib_of() { [ "$1" = default ] && echo $PWD/trueconsense_amd/lib/libtcmi.so || echo $PWD/trueconsense_amd/lib/var/lib$1.so; }
headline() {  # tag, env assignments..., -- bench args
  tag=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cli-batch --no-configs2 --no-configs0 --no-hard-bam --no-resident $BENCH_ARGS "$@" \
      > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err || { echo "$tag: fail"; tail -3 gpurun_out/ab/$tag.err; exit 1; }
  echo "$tag: $(python3 tools/bench_summary.py gpurun_out/ab/$tag.json 2>/dev/null | sed -n '1,3p' | tr '\n' ' ')"
}
case $mode in
tests)
  timeout -k 10 600 python -m pytest ${@:-tests/test_bam_device.py tests/test_bam_fixture.py tests/test_one_sync.py} -x -q -m gpu > gpurun_out/ab/tests.log 2>&1 || { tail -40 gpurun_out/ab/tests.log; exit 1; }
  tail -2 gpurun_out/ab/tests.log ;;
inflate|stamps)
  kinds=${1//,/ }; shift
  tool=inflate_time.py; pat='^(inflate|crc|counts)'; [ $mode = stamps ] && { tool=inflate_stamps.py; pat='.'; }
  for kind in $kinds; do for r in 1 2; do for v in "$@"; do
    echo "== $kind $v: $(TCMI_LIB=$(lib_of $v) timeout -k 10 300 python3 tools/$tool $kind 1000000 2>&1 | grep -v amdgpu.ids | grep -E "$pat" | tr '\n' ' ')"
    [ $mode = stamps ] && break
  done; [ $mode = stamps ] && break; done; done; true ;;
inflate-env)
  var=$1; kinds=${2//,/ }; shift 2
  for kind in $kinds; do for v in "$@"; do
    echo "== $kind $var=$v: $(env $var=$v timeout -k 10 300 python3 tools/inflate_time.py $kind 1000000 2>&1 | grep -E '^(inflate|crc|counts)' | tr '\n' ' ')"
  done; done ;;
bench)
  reps=$1; shift
  for r in $(seq $reps); do for v in "$@"; do headline $v TCMI_LIB=$(lib_of $v) -- ; done; done ;;
env)
  var=$1; reps=$2; shift 2
  for r in $(seq $reps); do for v in "$@"; do headline "$var=$v" $var=$v -- ; done; done ;;
ctxopt)
  key=$1; reps=$2; shift 2
  for r in $(seq $reps); do for v in "$@"; do headline "$key=$v" A=1 -- --ctx-option $key=$v; done; done ;;
args)
  reps=$1; shift
  for r in $(seq $reps); do for v in "$@"; do headline "$(echo $v | tr -c 'A-Za-z0-9\n' '_')" A=1 -- $v; done; done ;;
*) echo "usage: see the head of tools/gpu_ab.sh"; exit 2 ;;
esac
