#!/usr/bin/env python3
"""Decode the stderr timeline of a BERT_HIP_TIMELINE build of layer_tail (one stamp per tile interval): prints the
steady-state rows of 12 intervals = two chunk steps [UP UP UP DOWN DOWN DOWN] x 2 for the first dumped workgroup."""
import re, sys
for line in open(sys.argv[1]):
    m = re.match(r"timeline wg\s+(\d+):(.*)total (\d+)", line)
    if not m:
        continue
    d = [int(x) for x in m.group(2).split()]
    if len(d) < 150:
        continue
    body = d[18:170]
    print("wg", m.group(1), "prologue+PROJ", sum(d[:18]))
    for i in range(0, len(body), 12):
        print("  ", body[i:i + 12])
    break
