#!/bin/bash
export TMPDIR=/tmp
timeout 200 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -1
