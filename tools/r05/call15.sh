#!/bin/bash
VD3D_COMMIT=b127e0f bash tools/make_profiles.sh r05 2>&1 | tail -30
