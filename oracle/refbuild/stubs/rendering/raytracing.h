// Stand-in: scene.cpp includes rendering/raytracing.h for components the physics-only build never touches.
#pragma once
