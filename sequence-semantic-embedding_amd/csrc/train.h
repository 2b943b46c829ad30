// Training-path state (loss, BPTT, clip, Adagrad) -- see train.hip.
#pragma once
struct TrainState;
void train_state_free(TrainState *t);
