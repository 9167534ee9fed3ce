#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout ${TMO:-600} python3 -m pytest $TESTS -m gpu -x -q 2>&1 | tail -${TAIL:-6}
