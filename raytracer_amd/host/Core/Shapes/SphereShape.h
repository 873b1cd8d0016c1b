// reference include path compatibility: Core/Shapes/SphereShape.h
#pragma once
#include "Shape.h"
