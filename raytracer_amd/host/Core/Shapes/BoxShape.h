// reference include path compatibility: Core/Shapes/BoxShape.h
#pragma once
#include "Shape.h"
