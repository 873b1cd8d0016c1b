// reference include path compatibility: Core/Scene/Light/SpotLight.h
#pragma once
#include "Light.h"
