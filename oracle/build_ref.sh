#!/bin/bash
# oracle/build_ref.sh -- TEST INFRASTRUCTURE.
# Builds oracle/_ref/refdrv: our driver (oracle/ref_driver.cc) linked against the
# reference's own sources compiled WHERE THEY LIE under $REF (never copied, never edited).
# Flags only: -fpermissive -fno-access-control (gcc 4.8-era code under g++ 11),
# -include ref_compat.h (ifstream->bool), and clang -fdelayed-template-parsing for the one
# TU (Vec.cc) whose headers g++ 11 rejects.  Outputs only under oracle/_ref/ (git-ignored).
# Does nothing (exit 0) when the reference tree is absent (e.g. on the GPU box).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${DFK_REFERENCE:-/root/reference}/lib/assembly/src"
OUT="$HERE/_ref"
if [ ! -d "$REF" ]; then echo "build_ref: no reference tree at $REF; keeping prebuilt $OUT" ; exit 0; fi
mkdir -p "$OUT/obj"
CLANG=/opt/rocm/lib/llvm/bin/clang++
COMMON="-std=gnu++11 -O2 -w -fno-access-control -fopenmp -ffunction-sections -fdata-sections -include $HERE/ref_compat.h -I$REF"
cd "$REF"
LIST=$(ls system/*.cc system/file/*.cc feudal/*.cc dna/*.cc kmers/KMerContext.cc kmers/ReadPather.cc paths/long/ReadPath.cc 10X/Martian.cc \
          random/RNGen.cc math/PowerOf2.cc *.cc | grep -v -e MakeDepend.cc \
          -e '^Alignment.cc' -e BlockAlign.cc -e Fastavector.cc -e IndexedAlignmentPlusVector.cc \
          -e PackAlign.cc -e PrintAlignment.cc -e ScoreAlignment.cc -e VecAlignmentPlus.cc -e '^Vec.cc')
export REF OUT COMMON
echo "$LIST" | xargs -P "${DFK_JOBS:-8}" -I{} sh -c '
  o="$OUT/obj/$(echo {} | tr / _).o"
  if [ ! -f "$o" ] || [ "$REF/{}" -nt "$o" ]; then
    g++ $COMMON -fpermissive -c "{}" -o "$o" || { echo "build_ref: FAILED {}" >&2; exit 255; }
  fi'
o="$OUT/obj/Vec.cc.o"
if [ ! -f "$o" ]; then $CLANG $COMMON -fdelayed-template-parsing -c Vec.cc -o "$o"; fi
cd "$HERE"
# the graph half (digraphE<basevector>: graph/Digraph.h needs clang's delayed template parsing, like Vec.cc)
$CLANG $COMMON -fdelayed-template-parsing -c ref_graph.cc -o "$OUT/ref_graph.o"
g++ -no-pie $COMMON -fpermissive ref_driver.cc "$OUT/ref_graph.o" "$OUT"/obj/*.o -o "$OUT/refdrv" -Wl,--gc-sections -lz -lpthread
echo "build_ref: built $OUT/refdrv"
# The reference's own ParseBarcodedFastqs, from its own sources where they lie: 10X/ParseBarcodedFastqs.cc
# needs clang's delayed template parsing (kmers/KmerShape.h:566 under g++ 11) and is compiled WITHOUT -fopenmp: its
# only OpenMP loop appends the buckets' temporary files in the order the threads finish them, so the barcode order
# of a threaded run is not reproducible; single-threaded it is bucket 0, 1, 2, ...
cd "$REF"
$CLANG -std=gnu++11 -O2 -w -fno-access-control -include "$HERE/ref_compat.h" -I"$REF" -fdelayed-template-parsing -c 10X/ParseBarcodedFastqs.cc -o "$OUT/ParseBarcodedFastqs.o"
g++ $COMMON -fpermissive -c 10X/Barcode.cc -o "$OUT/Barcode.o"
cd "$HERE"
g++ -no-pie -fopenmp "$OUT/ParseBarcodedFastqs.o" "$OUT/Barcode.o" "$OUT"/obj/*.o -o "$OUT/ParseBarcodedFastqs" -Wl,--gc-sections -lz -lpthread
echo "build_ref: built $OUT/ParseBarcodedFastqs"
