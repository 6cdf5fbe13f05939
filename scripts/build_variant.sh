#!/bin/bash
# dev: build the library with extra compile flags into ab/<name>.so (for scripts/ab.sh), then
# (afterwards `python -m thrifty_amd.build` restores the default: the flag stamp forces the rebuild).   scripts/build_variant.sh dev -DTHR_DEV
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p ab
THR_EXTRA_CFLAGS="$*" python -m thrifty_amd.build --force > /dev/null
cp thrifty_amd/libthriftyhip.so ab/$NAME.so
echo "ab/$NAME.so  [$*]"
