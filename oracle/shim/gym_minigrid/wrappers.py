"""Restatement of the gym_minigrid.wrappers the reference uses
(babyai/evaluate.py:91-92, scripts/train_rl.py:57-58).  TEST INFRASTRUCTURE."""
import gym
from gym import spaces


class ImgObsWrapper(gym.core.ObservationWrapper):
    """Use the image as the only observation output, no language/mission."""

    def __init__(self, env):
        super().__init__(env)
        self.observation_space = env.observation_space.spaces['image']

    def observation(self, obs):
        return obs['image']


class RGBImgPartialObsWrapper(gym.core.ObservationWrapper):
    """Partially observable (egocentric 7x7 view) RGB image as observation;
    tile_size=8 => uint8[56,56,3] (row = view y, col = view x)."""

    def __init__(self, env, tile_size=8):
        super().__init__(env)
        self.tile_size = tile_size
        obs_shape = env.observation_space.spaces['image'].shape
        self.observation_space = spaces.Dict({
            'image': spaces.Box(
                low=0, high=255,
                shape=(obs_shape[0] * tile_size, obs_shape[1] * tile_size, 3),
                dtype='uint8'),
        })

    def observation(self, obs):
        env = self.unwrapped
        rgb_img_partial = env.get_obs_render(obs['image'], tile_size=self.tile_size)
        return {'mission': obs['mission'], 'image': rgb_img_partial}


class FullyObsWrapper(gym.core.ObservationWrapper):
    """Fully observable grid encoding (not used on the hot path)."""

    def __init__(self, env):
        super().__init__(env)
        self.observation_space = spaces.Dict({
            'image': spaces.Box(low=0, high=255,
                                shape=(self.env.width, self.env.height, 3), dtype='uint8'),
        })

    def observation(self, obs):
        import numpy as np
        from .minigrid import COLOR_TO_IDX, OBJECT_TO_IDX
        env = self.unwrapped
        full_grid = env.grid.encode()
        full_grid[env.agent_pos[0]][env.agent_pos[1]] = np.array(
            [OBJECT_TO_IDX['agent'], COLOR_TO_IDX['red'], env.agent_dir])
        return {'mission': obs['mission'], 'image': full_grid}
