#!/bin/bash
# builds the tools-only probe libraries (gfx950) next to their sources
set -e
cd "$(dirname "$0")"
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$f" -o "${f%.hip}.so"
done
