// reference include path compatibility: Core/Shapes/RectShape.h
#pragma once
#include "Shape.h"
