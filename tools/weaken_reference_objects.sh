#!/bin/bash
# Link-time swap of the matcher bodies, no source edit in the reference tree:
#   the reference's src/Frame.cc and src/ORBmatcher.cc are compiled UNTOUCHED; this script then marks, in those two object files,
#   the six member functions adapter/matchers_gfo.cc re-implements as WEAK definitions.  At link time the adapter's (strong)
#   definitions win -- for every caller, including the calls inside Frame.o / ORBmatcher.o themselves (Frame::Frame ->
#   ComputeStereoMatches_Undistorted, Frame.cc:100) -- and the reference's bodies stay in the binary, unreachable.
# usage: tools/weaken_reference_objects.sh <Frame.cc.o> <ORBmatcher.cc.o> [more objects]
# CMake (reference's CMakeLists.txt, after add_library(${PROJECT_NAME} SHARED ${SRCS} <adapter files>)):
#   add_custom_command(TARGET ${PROJECT_NAME} PRE_LINK
#       COMMAND /path/to/repo/tools/weaken_reference_objects.sh
#               ${CMAKE_CURRENT_BINARY_DIR}/CMakeFiles/${PROJECT_NAME}.dir/src/Frame.cc.o
#               ${CMAKE_CURRENT_BINARY_DIR}/CMakeFiles/${PROJECT_NAME}.dir/src/ORBmatcher.cc.o)
set -e
LIST="$(dirname "$(readlink -f "$0")")/../gf-orb-slam2_amd/adapter/weaken_symbols.txt"
OBJCOPY=${OBJCOPY:-objcopy}
for o in "$@"; do
  "$OBJCOPY" --weaken-symbols="$LIST" "$o"
done
