class DQN:  # placeholder until xq_dqn lands
    pass
class Trainer:
    pass
class TrainerConfig:
    pass
