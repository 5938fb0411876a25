set -e
bash tools/profile_round.sh r02f "mixed sine_f32" BDI
bash tools/profile_round.sh r02f "random_u32" BPC
