char __hip_fatbin_a141d348890333b4[16] = {0};
