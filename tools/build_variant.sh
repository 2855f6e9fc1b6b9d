#!/bin/bash
# tools/build_variant.sh NAME "FLAGS" — a build of libtcmi.so with extra compiler flags as trueconsense_amd/lib/var/libNAME.so (objects under
# var/objNAME), for the A/B runs of tools/gpu_ab.sh; e.g.  tools/build_variant.sh ring4k "-DTCMI_COPY_RING=4096 -DTCMI_COPY_SEG=1024"
set -e
cd "$(dirname "$0")/../trueconsense_amd/csrc"
make -s -j8 OUT=../lib/var/lib$1.so OBJDIR=../lib/var/obj$1 RCCL_OUT= EXTRA="$2"      # (RCCL_OUT empty: a variant has no hook library of its own)
ls -la ../lib/var/lib$1.so
