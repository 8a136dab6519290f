#!/bin/bash
# Development: an A/B build of the library with extra compiler flags -> tools/bin/libaie_<name>.so (loaded through
# AIE_HIP_LIBRARY by tools/ab_variants.sh).   tools/build_variant.sh <name> [-DFLAG ...]
R=$(cd $(dirname $0)/.. && pwd)
NAME=$1; shift
mkdir -p $R/tools/bin
python - <<PY
import sys
sys.path.insert(0, "$R")
import ai_economist_amd
from ai_economist_amd import _build
_build.build()  # (the version script beside the shipping library)
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -fvisibility=hidden \
  -Wl,--version-script=$R/ai-economist_amd/csrc/libaie_hip.so.map -Wno-comment -I$R/include "$@" \
  $R/ai-economist_amd/csrc/aie_capi.hip -o $R/tools/bin/libaie_$NAME.so && echo built tools/bin/libaie_$NAME.so
