#!/bin/bash
# tools/whatif_build.sh <patch under experiments/> <variant>: a what-if build of the library
# (WRONG results by design: timing only) from a patched scratch copy of csrc/ -- the product
# tree is not touched.  -> extensisq_amd/libextensisq_amd_<variant>.so  (load with ESQ_LIB=)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
patch=$1; variant=$2
[ -f "$root/experiments/$patch" ] && [ -n "$variant" ] || { echo "usage: $0 <patch> <variant>"; exit 2; }
tmp=$(mktemp -d /tmp/esq_whatif.XXXXXX)
mkdir -p "$tmp/extensisq_amd" "$tmp/include"
cp -r "$root/extensisq_amd/csrc" "$tmp/extensisq_amd/csrc"
cp "$root/include/extensisq_amd.h" "$tmp/include/"
rm -rf "$tmp/extensisq_amd/csrc/build" "$tmp/extensisq_amd/csrc"/build_*
(cd "$tmp" && patch -p1 < "$root/experiments/$patch")
make -j8 -C "$tmp/extensisq_amd/csrc" >/dev/null
cp "$tmp/extensisq_amd/libextensisq_amd.so" "$root/extensisq_amd/libextensisq_amd_$variant.so"
rm -rf "$tmp"
echo "built extensisq_amd/libextensisq_amd_$variant.so"
