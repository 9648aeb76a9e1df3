from .loader import RobotModel, build_robot_model, load_robot_model, self_collision_pairs  # noqa: F401
