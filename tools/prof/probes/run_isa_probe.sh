#!/bin/bash
# every test in its own process with its own limit: a test that hangs costs 15 s, not the session
P=$(dirname $0)/isa_rate_probe
for i in $(seq 0 56); do timeout 15 $P $i $((i+1)) | grep -v "^device" ; rc=${PIPESTATUS[0]}; [ $rc -ne 0 ] && echo "[$i] rc=$rc"; done
