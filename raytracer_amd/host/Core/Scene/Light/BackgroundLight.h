// reference include path compatibility: Core/Scene/Light/BackgroundLight.h
#pragma once
#include "Light.h"
