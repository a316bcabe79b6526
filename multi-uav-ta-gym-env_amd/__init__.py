"""MI355X-native batched mUAV_TA environment."""
