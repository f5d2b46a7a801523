export LAE_FRAME_LOOK_EARLY=1
bash tools/r4_frame_profile.sh fl1e
