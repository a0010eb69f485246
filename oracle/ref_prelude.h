// ref_prelude.h — TEST INFRASTRUCTURE.  Force-included ahead of the reference translation unit
// in the `ref_render_depth` build only (oracle/Makefile).  That build's sed rewrites the literal
// local `int maxBounces = 1;` (reference Source.cpp:567) to read this variable so the reference's
// own bounce loop (Source.cpp:570-594) can be run at depths other than 1.  Default 1 == reference.
#pragma once
static int g_oracle_max_bounces = 1;
