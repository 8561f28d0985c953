#!/bin/bash
cd "$(dirname "$0")/../../.."
while read -r n cfg; do echo "=== $cfg"; timeout 120 tools/ubench/msda_mm/bin/mm_$n; done < tools/ubench/msda_mm/bin/list.txt
