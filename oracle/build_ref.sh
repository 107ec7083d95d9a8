#!/bin/sh
# oracle/build_ref.sh <reference root> -- placeholder until the host-compiled reference device
# functions land (see oracle/ref_shim/).  Outputs only into oracle/_ref/.
set -e
mkdir -p "$(dirname "$0")/_ref"
exit 0
