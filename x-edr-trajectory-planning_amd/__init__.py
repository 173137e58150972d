"""MI355X-native batched time-optimal path timing."""
