// reference include path compatibility: Core/Scene/Light/AreaLight.h
#pragma once
#include "Light.h"
