// pcl/registration/correspondence_rejection_surface_normal.h - see correspondence_rejection.h
#include <pcl/registration/correspondence_rejection.h>
