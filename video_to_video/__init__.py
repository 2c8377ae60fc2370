"""Drop-in package path of the reference (`video_to_video/`): thin re-exports of the MI355X implementation in star_amd."""
