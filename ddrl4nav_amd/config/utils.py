"""Environment-name -> task-type table (mirror of USTC_lab/config/utils.py:17-36; only the
families the hot path serves are distinguished, the rest map to their reference names)."""
_ATARI = ("AirRaid Alien Amidar Assault Asterix Asteroids Atlantis BankHeist BattleZone BeamRider Berzerk Bowling "
          "Boxing Breakout Carnival Centipede ChopperCommand CrazyClimber DemonAttack DoubleDunk ElevatorAction "
          "Enduro FishingDerby Freeway Frostbite Gopher Gravitar IceHockey Jamesbond JourneyEscape Kangaroo Krull "
          "KungFuMaster MontezumaRevenge MsPacman NameThisGame Phoenix Pitfall Pong Pooyan PrivateEye Qbert "
          "Riverraid RoadRunner Robotank Seaquest Skiing Solaris SpaceInvaders StarGunner Tennis TimePilot "
          "Tutankham UpNDown Venture VideoPinball WizardOfWor YarsRevenge Zaxxon").split()
_GROUPS = [
    ("atari", _ATARI),
    ("classical", ["Acrobot", "CartPole", "MountainCar", "MountainCarContinuous", "Pendulum"]),
    ("mujoco", ["Ant", "HalfCheetah", "Hopper", "Humanoid", "HumanoidStandup", "InvertedDoublePendulum",
                "InvertedPendulum", "Reacher", "Swimmer", "Walker2d"]),
    ("robot_nav", ["Navigation", "robotnav", "ROBOTnav", "robot_nav"]),
]


def startswith_groups(x, groups):
    return any(x.startswith(y) for y in groups)


def game_type(gid):
    for name, groups in _GROUPS:
        if startswith_groups(gid, groups):
            return name
    raise NameError(gid)
