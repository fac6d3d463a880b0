"""Yaw-only quaternion helpers with the (x, y, z, w) layout ROS uses."""
import math


def quaternion_from_euler(roll, pitch, yaw, axes="sxyz"):
    if roll or pitch:
        raise NotImplementedError("planar robots: only yaw is supported")
    return [0.0, 0.0, math.sin(0.5 * yaw), math.cos(0.5 * yaw)]


def euler_from_quaternion(q, axes="sxyz"):
    x, y, z, w = q
    return (0.0, 0.0, math.atan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z)))
