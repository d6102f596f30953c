#!/bin/bash
# dev tool: build libbioscan.so with extra compile flags into tools/_build/variants/NAME/ (travels to the GPU box;
# tools/k1_variants.sh swaps it in there).  usage: build_variant.sh NAME "-DV3_SUB_DW=104 ..."
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
V=$R/tools/_build/variants/$1
mkdir -p $V/csrc
ln -sfn ../../../include $R/tools/_build/variants/include
cp $R/datafusion-bio-formats_amd/csrc/*.hip $R/datafusion-bio-formats_amd/csrc/*.cpp $R/datafusion-bio-formats_amd/csrc/*.h $R/datafusion-bio-formats_amd/csrc/*.inc $R/datafusion-bio-formats_amd/csrc/Makefile $V/csrc/
make -C $V/csrc -j8 EXTRA="$2" OUT=$V/libbioscan.so 2>&1 | grep -E "error|Error" || true
ls -la $V/libbioscan.so
