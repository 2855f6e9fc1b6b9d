#!/bin/bash
# bgzf_symbols with the payload staged whole (TCMI_SYM_WINDOW=0) or a window at a time (bytes), on files of two compression ratios
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_bam_device.py tests/test_bam_fixture.py -x -q -m gpu 2>&1 | tail -2 || exit 1
for kind in hard real; do
  for w in 0 4096 6144 8192; do
    echo "== $kind window=$w: $(TCMI_SYM_WINDOW=$w timeout -k 10 300 python3 tools/inflate_stamps.py $kind 1000000 2>&1 | grep -E 'inflate\(symbols\)|inflate\(copy\)|counts equal|tcmi inflate\]|workgroup total' | sort | uniq | tr '\n' '|' | cut -c1-560)"
  done
done
