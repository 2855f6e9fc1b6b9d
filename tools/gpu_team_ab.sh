#!/bin/bash
# bgzf_copy: the batch size (bytes of output per 64 tokens) up to which matches are copied in teams, on the bench file and the harder one
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tb in 0 1024 1536 2560 4096 100000; do
  for kind in headline hard; do
    echo "team_bytes $tb $kind: $(TCMI_TEAM_BYTES=$tb timeout -k 10 300 python3 tools/inflate_stamps.py $kind 1000000 2>&1 | grep -E 'inflate\(copy\)|copy total' | tr '\n' ' ')"
  done
done
