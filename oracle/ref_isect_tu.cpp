/*
 * Translation unit that compiles the reference's src/intersections.cu, unmodified, as host C++ for
 * oracle/_ref/ref_tool.  One accommodation: sphereIntersectionTest calls the unqualified float
 * min / max (src/intersections.cu:95, :100) that nvcc's front end declares globally; a host C++
 * compiler has no such declaration, so the standard ones are made visible here, before the
 * reference's file is read.  For the non-NaN operands that reach those two calls (the radicand
 * test above them returns first) std::min / std::max and CUDA's fminf / fmaxf forms agree.
 * Test infrastructure only.
 */
#include <algorithm>
#include <cmath>
using std::max;
using std::min;
#include "intersections.cu"
