// reference include path compatibility: Core/Scene/Light/PointLight.h
#pragma once
#include "Light.h"
