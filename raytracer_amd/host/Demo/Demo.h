// The slice of the Demo's global options that the loaders read (reference: Demo/Demo.h -- Options::dataPath is
// prepended to every mesh / texture path of a scene file, SceneLoader.cpp:237, 311, 447).
#pragma once

#include <string>

#include "../Core/Math/Math.h"   // RAYLIB_API

struct Options
{
    std::string dataPath;
};

extern RAYLIB_API Options gOptions;
