// reference include path compatibility: Core/Shapes/MeshShape.h
#pragma once
#include "Shape.h"
