#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY -- never used by the product path.
#
# Compiles the *unmodified* reference sources where they lie (/root/reference/{calc,hitPosition}.cpp
# plus the headers they include) together with oracle/ref_harness.cpp into oracle/_ref/ref_harness.
# Recipe follows SURVEY.md section 8c: g++ + the Qt5 (Core, Gui) already installed in this image
# under /opt/conda; no reference build system is run, no reference source is copied or edited.
# MSVC-isms are handled with compiler flags only:
#   -include memory/vector/algorithm/optional : headers calc.h forgets (MSVC pulls them transitively)
#   -fpermissive                              : sphere::material member shadows class material
#   -fno-access-control                       : lets the harness read camera's private members
# Deliberately NOT used: -march=native / -ffast-math (baseline x86-64 has no FMA; the reference
# arithmetic contract is separately rounded FP32 + FP64 libm scalars).
#
# oracle/_ref/ is git-ignored (binary only) but NOT gpurun-ignored.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
REF="${RT_REFERENCE_DIR:-/root/reference}"
Q="${RT_QT_INCLUDE:-/opt/conda/include/qt}"
QL="${RT_QT_LIB:-/opt/conda/lib}"
if [ ! -f "$REF/calc.cpp" ] || [ ! -d "$Q/QtGui" ]; then
    echo "build_ref.sh: reference sources or Qt5 headers not present; skipping (_ref not built)" >&2
    exit 0
fi
mkdir -p "$here/_ref"
# the mangled name of hitFromList, for the ray counter of ref_harness.cpp (linker --wrap: no reference source is touched)
g++ -std=c++17 -O2 -fPIC -fpermissive -w -ffp-contract=off -include memory -include vector -include algorithm -include optional \
    -I"$REF" -I"$Q" -I"$Q/QtCore" -I"$Q/QtGui" -c "$REF/hitPosition.cpp" -o "$here/_ref/hitPosition.probe.o"
sym="$(nm "$here/_ref/hitPosition.probe.o" | awk '$2 == "T" && $3 ~ /hitFromList/ { print $3; exit }')"
rm -f "$here/_ref/hitPosition.probe.o"
[ -n "$sym" ] || { echo "build_ref.sh: hitFromList not found in hitPosition.o" >&2; exit 1; }
g++ -std=c++17 -O2 -fPIC -fpermissive -fno-access-control -w -ffp-contract=off \
    -include memory -include vector -include algorithm -include optional \
    -I"$REF" -I"$Q" -I"$Q/QtCore" -I"$Q/QtGui" \
    "$here/ref_harness.cpp" "$REF/calc.cpp" "$REF/hitPosition.cpp" \
    -DRT_HFL_SYM="\"$sym\"" -Wl,--wrap="$sym" \
    -L"$QL" -lQt5Gui -lQt5Core -Wl,-rpath,"$QL" \
    -o "$here/_ref/ref_harness"
echo "built $here/_ref/ref_harness"
