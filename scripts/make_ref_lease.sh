#!/bin/bash
# Pack the reference's DRIVER files (train.py, opt.py, gui.py, datasets/ -- never modules/) into ref_lease.tgz at the repo root so
# that a gpurun call can run the unchanged driver on the GPU box (scripts/run_reference_train.py), where /root/reference does
# not exist.  The tarball is git-ignored and must be removed after the call: reference sources never enter this repo's history.
#   scripts/make_ref_lease.sh && gpurun ... ; scripts/make_ref_lease.sh --remove
set -euo pipefail
cd "$(dirname "$0")/.."
if [ "${1:-}" = "--remove" ]; then rm -f ref_lease.tgz; exit 0; fi
REF=${REF:-/root/reference}
tar czf ref_lease.tgz --exclude=__pycache__ -C "$REF" train.py opt.py gui.py datasets
ls -la ref_lease.tgz
