// "Path Tracer" (reference: Core/Rendering/PathTracer.h): the BSDF-sampling walk without next event estimation.  The class lives next to
// PathTracerMIS, whose device pipeline it shares; this header exists so that callers written against the reference's include paths
// (Tests/RaytracingTests.cpp:7) build unchanged.
#pragma once

#include "PathTracerMIS.h"
