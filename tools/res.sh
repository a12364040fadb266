#!/bin/bash
# Register, scratch, LDS and occupancy of every kernel, one line each.
cd "$(dirname "$0")/.." && make resources DEFS="$*" 2>/dev/null | sed -e 's/\[-Rpass-analysis=kernel-resource-usage\]//g' -e 's/Function Name: //' | awk '{print}' | sed -e 's/  */ /g' | cut -c1-200
