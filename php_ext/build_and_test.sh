#!/usr/bin/env bash
# Builds the Zend shim (php_ext/php_ahocorasick_amd.c) against this repo's libacmi.so and runs the REFERENCE's own .phpt
# files -- plus this repo's php_ext/tests/*.phpt -- through it.
#
#     php_ext/build_and_test.sh /path/to/php_aho_corasick   # a checkout of ph4r05/php_aho_corasick (its tests/ are used in place)
#
# Needs php-dev (phpize, php-config), a C compiler and an MI355X (gfx950) -- there is no CPU matching path.  NOT RUN in
# this repo's build image: it has no PHP toolchain (no php, phpize, php-config or Zend headers), so the shim has never
# been compiled; this script is what a maintainer with php-dev runs.  The reference's .phpt files are not copied: they are
# read from the checkout given on the command line.
set -euo pipefail
REF="${1:?usage: build_and_test.sh /path/to/reference/checkout}"
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REPO="$(dirname "$HERE")"
for tool in phpize php-config php; do
    command -v "$tool" >/dev/null || { echo "$tool not found: install php-dev / php-cli" >&2; exit 2; }
done
[ -d "$REF/tests" ] || { echo "$REF/tests not found: give the reference checkout" >&2; exit 2; }

# 1. libacmi.so (hipcc, gfx950)
python3 -m php_aho_corasick_amd.build >/dev/null 2>&1 || (cd "$REPO" && python3 -m php_aho_corasick_amd.build)
[ -f "$REPO/php_aho_corasick_amd/libacmi.so" ] || { echo "libacmi.so missing" >&2; exit 2; }

# 2. the extension, out of tree
BUILD="$(mktemp -d)"
cp "$HERE/config.m4" "$HERE/php_ahocorasick_amd.c" "$BUILD/"
cd "$BUILD"
phpize
./configure --enable-ahocorasick --with-acmi="$REPO"
make -j"$(nproc)"

# 3. the reference's tests and ours through the built module
mkdir -p "$BUILD/tests"
cp "$REF"/tests/*.phpt "$BUILD/tests/"          # (into the scratch build dir only)
cp "$HERE"/tests/*.phpt "$BUILD/tests/"
export LD_LIBRARY_PATH="$REPO/php_aho_corasick_amd:${LD_LIBRARY_PATH:-}"
export NO_INTERACTION=1 REPORT_EXIT_STATUS=1
export TEST_PHP_ARGS="-q"
php run-tests.php -d extension="$BUILD/modules/ahocorasick.so" -q tests/
